"""saveParticleData / restoreParticleData — the reference's text checkpoint (utils/checkpoint.h:29-76), the format a real
UAMMD run writes and reads, kept byte-compatible so that states can be exchanged with an NVIDIA box:

    # version 3.0.0
    # <numberParticles>
    # <Name>                 one block per ALLOCATED property, in the order of ParticleData's property list
    <v0> [<v1> <v2> <v3>]    N lines in particle-ID order (v[id2index[i]]), C++ default stream formatting (%g, 6 digits)

Id itself is never written (checkpoint.h:17-18); the reader creates ids 0..N-1 and fills the properties it finds.
"""
import numpy as np
import torch

from .md import ParticleData

UAMMD_VERSION = "3.0.0"  # global/defines.h:7
# ParticleData.cuh:33-46 (AngVel is not a property of this build's ParticleData; a block of that name is skipped on read)
_PROPERTIES = [("Pos", "pos", 4), ("Mass", "mass", 1), ("Force", "force", 4), ("Virial", "virial", 1), ("Energy", "energy", 1),
               ("Vel", "vel", 3), ("Radius", "radius", 1), ("Charge", "charge", 1), ("Torque", "torque", 4), ("AngVel", None, 4),
               ("Dir", "dir", 4)]


def _fmt(x, precision):
    return ("%." + str(precision) + "g") % float(x)


def saveParticleData(fileName, pd, precision=6):
    """precision = 6 is what `out << value` gives in the reference; a larger value writes a lossless file the reference
    reader accepts just the same."""
    ids = pd.id.cpu().numpy()
    id2index = np.empty(pd.N, np.int64)
    id2index[ids] = np.arange(pd.N)          # ParticleData::getIdOrderedIndices
    with open(fileName, "w") as out:
        out.write("# version %s\n" % UAMMD_VERSION)
        out.write("# %d\n" % pd.N)
        for Name, name, width in _PROPERTIES:
            if name is None or not pd.isAllocated(name):
                continue
            v = pd._props[name].cpu().numpy().reshape(pd.N, -1)[id2index]
            out.write("# %s\n" % Name)
            for row in v:
                out.write(" ".join(_fmt(x, precision) for x in row) + "\n")


def restoreParticleData(fileName, device="cuda"):
    with open(fileName) as f:
        tok = f.read().split()
    # "#" "version" <ver> "#" <N>, then repeated "#" <Name> <values...>
    assert tok[0] == "#" and tok[1] == "version", "not a UAMMD checkpoint"
    n = int(tok[4])
    pd = ParticleData(n, device=device)
    widths = {Name: (name, width) for Name, name, width in _PROPERTIES}
    i = 5
    while i < len(tok):
        assert tok[i] == "#", "malformed checkpoint block header"
        Name = tok[i + 1]
        i += 2
        if Name not in widths:
            raise ValueError("unknown property %s in checkpoint" % Name)
        name, width = widths[Name]
        vals = np.array(tok[i:i + n * width], dtype=np.float32).reshape(n, width)
        i += n * width
        if name is None:
            continue
        t = pd._get(name, width)
        t.copy_(torch.from_numpy(vals if width > 1 else vals[:, 0]).to(t.device))
    return pd
