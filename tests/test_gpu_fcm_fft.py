"""The solver's own mixed-radix FFT pipeline (csrc/fcm_fft.hpp: axis lengths 2^a 3^b 5^c 7^d 11^e — what Grid's nextFFTWiseSize3D hands out) against rocFFT + the stand-alone Fourier-space kernel on the same handle type, and against the oracle."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GRIDS = [(54, 54, 54), (48, 40, 60), (96, 80, 50), (108, 108, 108), (30, 36, 20), (20, 12, 16), (120, 90, 150), (162, 128, 100),
         (250, 256, 12), (64, 100, 27), (32, 16, 500), (18, 250, 8), (128, 128, 128), (36, 30, 25),
         (112, 84, 98), (110, 66, 44), (28, 22, 14), (154, 126, 242), (56, 121, 49),
         # the passes fused with the loads / stores: rows starting with radix 4 and radix 8, columns of one pass, z lines of 32 and 256
         (32, 32, 32), (16, 8, 64), (64, 64, 256), (256, 32, 128)]


def _inputs(cells, n, seed):
    L = np.asarray(cells, np.float32)
    rng = np.random.default_rng(seed)
    pos = np.zeros((n, 4), np.float32)
    pos[:, :3] = rng.uniform(-0.7, 0.7, (n, 3)) * L
    force = np.zeros((n, 4), np.float32)
    force[:, :3] = rng.normal(0, 1, (n, 3))
    return L, pos, force


@pytest.mark.parametrize("cells", GRIDS, ids=["x".join(map(str, g)) for g in GRIDS])
def test_fcm_own_fft_matches_rocfft(hip, cells):
    """Velocities with the five-pass pipeline (rows / planes, y lines, fused z + operator, y lines, rows) and with rocFFT's batched 3-D
    transforms around k_fcm_kspace: T = 0 and T > 0 (same seeds, same noise call number: the same noise field node by node)."""
    n = 3000
    L, pos, force = _inputs(cells, n, sum(cells))
    k, a_eff = hip.Kernels.Gaussian(1.0, 1e-3)
    dp, df = torch.from_numpy(pos).cuda(), torch.from_numpy(force).cuda()
    out = {}
    for custom in (1, 0):
        fcm = hip.BDHI.FCM_impl(hip.Box(L), list(cells), k, 0.9, 5, a_eff)
        fcm.set_option("custom_fft", custom)
        v0 = fcm.computeHydrodynamicDisplacements(dp, df, n, 0.0, 0.0).cpu().numpy()
        v1 = fcm.computeHydrodynamicDisplacements(dp, df, n, 0.7, 2.0).cpu().numpy()
        v2 = fcm.computeHydrodynamicDisplacements(dp, None, n, 0.7, 2.0).cpu().numpy()   # noise only
        out[custom] = (v0, v1, v2)
    for a, b in zip(out[1], out[0]):
        assert np.isfinite(a).all() and np.abs(b).max() > 0
        assert np.linalg.norm(a - b) <= 1e-5 * np.linalg.norm(b)


@pytest.mark.parametrize("cells", [(54, 54, 54), (30, 36, 20), (48, 40, 60), (28, 22, 42)], ids=["54", "30x36x20", "48x40x60", "28x22x42"])
def test_fcm_own_fft_vs_oracle(hip, o32, cells):
    from oracle.fcm import FCMOracle
    n = 500
    L, pos, force = _inputs(cells, n, 7 + sum(cells))
    k, a_eff = hip.Kernels.Gaussian(1.0, 1e-3)
    fcm = hip.BDHI.FCM_impl(hip.Box(L), list(cells), k, 1.1, 77, a_eff)
    ofcm = FCMOracle(o32, L, list(cells), tolerance=1e-3, viscosity=1.1, seed=77)
    dp, df = torch.from_numpy(pos).cuda(), torch.from_numpy(force).cuda()
    v = fcm.computeHydrodynamicDisplacements(dp, df, n, 0.0, 0.0).cpu().numpy()
    vref = ofcm.displacements(pos, force)
    assert np.linalg.norm(v - vref) <= 1e-5 * np.linalg.norm(vref)
    T, dt = 0.8, 0.01
    v = fcm.computeHydrodynamicDisplacements(dp, df, n, T, 1 / math.sqrt(dt)).cpu().numpy()
    vref = ofcm.displacements(pos, force, temperature=T, prefactor=1 / math.sqrt(dt))
    assert np.linalg.norm(v - vref) <= 2e-5 * np.linalg.norm(vref)
