

def test_signal_connections_do_not_keep_receivers_alive():
    """ParticleData's write / reorder signals hold bound methods weakly (the reference's receivers own connection objects that disconnect
    with them): a solver is freed when its owner drops it — not when the cyclic collector runs, possibly inside somebody's timed loop with
    a device-synchronising free per buffer — and a dead receiver is dropped from the list at the next emission."""
    import gc
    import weakref
    import uammd_amd as hip
    pd = hip.ParticleData(8, seed=1, device="cpu")
    calls = []

    class Receiver:
        def __init__(self, pd):
            self.pd = pd
            pd.connectPosWrite(self.on_write)
            pd.connectReorder(self.on_write)

        def on_write(self):
            calls.append(1)
    gc.disable()
    try:
        r = Receiver(pd)
        pd.getPos("write")
        assert calls == [1]
        ref = weakref.ref(r)
        del r
        assert ref() is None                     # freed by its reference count: no cycle through the signal
        pd.getPos("write")
        assert calls == [1] and pd._pos_write_callbacks == []
        keep = []
        pd.connectPosWrite(lambda: keep.append(1))   # a plain function is held as it is
        pd.getPos("write")
        assert keep == [1]
    finally:
        gc.enable()
