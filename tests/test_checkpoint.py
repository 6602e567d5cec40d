"""The text checkpoint of utils/checkpoint.h (saveParticleData / restoreParticleData): the data format on the far side of
ParticleData.  tests/golden/checkpoint_example.dat is the file the reference's examples/misc/checkpoint.cu writes (N = 10,
Pos = 1, Charge = 2; derived by hand from checkpoint.h:29-52 and printOverloads.h:9-19 — the reference cannot run here)."""
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden", "checkpoint_example.dat")


def _pd(n):
    from uammd_amd.md import ParticleData
    return ParticleData(n, device="cpu")


def test_reference_example_file_roundtrip(tmp_path):
    from uammd_amd.checkpoint import restoreParticleData, saveParticleData
    pd = _pd(10)
    pd.getPos("write").fill_(1.0)
    pd.getCharge("write").fill_(2.0)
    out = tmp_path / "pd.dat"
    saveParticleData(str(out), pd)
    assert out.read_text() == open(GOLD).read()          # byte for byte what UAMMD writes
    back = restoreParticleData(GOLD, device="cpu")
    assert back.N == 10 and torch.equal(back.getPos("read"), pd.getPos("read")) and torch.equal(back.getCharge("read"), pd.getCharge("read"))
    assert not back.isAllocated("vel") and not back.isAllocated("force")   # only the blocks in the file


def test_id_order_and_every_property(tmp_path):
    """Values are written in particle-ID order whatever the memory order (v[id2index[i]]); Id itself is not written."""
    from uammd_amd.checkpoint import restoreParticleData, saveParticleData
    n = 37
    rng = np.random.default_rng(0)
    pd = _pd(n)
    perm = torch.from_numpy(rng.permutation(n).astype(np.int32))
    pd.id = perm.clone()                                  # memory slot k holds particle perm[k], as after sortParticles
    byid = {}
    for name, width, getter in [("pos", 4, pd.getPos), ("vel", 3, pd.getVel), ("mass", 1, pd.getMass), ("charge", 1, pd.getCharge),
                                ("dir", 4, pd.getDir), ("torque", 4, pd.getTorque), ("energy", 1, pd.getEnergy)]:
        vals = rng.normal(0, 10, (n, width)).astype(np.float32)   # indexed by particle id
        byid[name] = vals
        t = getter("write")
        t.copy_(torch.from_numpy(vals[perm.numpy()] if width > 1 else vals[perm.numpy(), 0]))
    f = tmp_path / "all.dat"
    saveParticleData(str(f), pd, precision=9)             # lossless for float32; the block layout is the same
    text = f.read_text().split("\n")
    assert text[0] == "# version 3.0.0" and text[1] == "# %d" % n and "# Id" not in text
    heads = [l for l in text if l.startswith("# ") and not l[2].isdigit() and not l.startswith("# version")]
    assert heads == ["# Pos", "# Mass", "# Energy", "# Vel", "# Charge", "# Torque", "# Dir"]   # ParticleData.cuh:33-46 order
    back = restoreParticleData(str(f), device="cpu")
    assert torch.equal(back.id, torch.arange(n, dtype=torch.int32))
    for name, vals in byid.items():
        got = back._props[name].numpy().reshape(n, -1)
        assert np.array_equal(got, vals), name
    # default precision = the reference's `out << value`: 6 significant digits
    g = tmp_path / "six.dat"
    saveParticleData(str(g), pd)
    back6 = restoreParticleData(str(g), device="cpu")
    assert np.abs(back6.getPos("read").numpy() - byid["pos"]).max() <= 5e-6 * np.abs(byid["pos"]).max()


def test_unknown_blocks(tmp_path):
    from uammd_amd.checkpoint import restoreParticleData
    f = tmp_path / "angvel.dat"
    f.write_text("# version 2.9.9\n# 2\n# Pos\n0 0 0 0\n1 2 3 4\n# AngVel\n0 0 1 0\n0 1 0 0\n# Charge\n-1\n1\n")
    pd = restoreParticleData(str(f), device="cpu")       # AngVel is a reference property this build does not carry: skipped
    assert pd.getCharge("read").tolist() == [-1.0, 1.0] and pd.getPos("read")[1].tolist() == [1.0, 2.0, 3.0, 4.0]
