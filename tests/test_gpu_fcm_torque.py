"""GPU parity: FCM torques / rotation (uammd_fcm_displacements_torque, uammd_fcm_euler_maruyama_dir) vs the oracle, and the
physics the reference has no test for: the rotational mobility of an isolated FCM particle, omega = tau / (8 pi eta a^3)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_torque_displacements_match_oracle(hip, o32):
    from oracle.fcm import FCMOracle
    cells, L, n, visc, seed = [48, 48, 48], 48.0, 1500, 1.2, 77
    rng = np.random.default_rng(3)
    pos = np.zeros((n, 4), np.float32)
    pos[:, :3] = rng.uniform(-L / 2, L / 2, (n, 3))
    force = np.zeros((n, 4), np.float32)
    force[:, :3] = rng.normal(0, 1, (n, 3))
    torque = np.zeros((n, 4), np.float32)
    torque[:, :3] = rng.normal(0, 1, (n, 3))
    k, a = hip.Kernels.Gaussian(1.0, 1e-3)
    fcm = hip.BDHI.FCM_impl(hip.Box(L), cells, k, visc, seed, a)
    kt = fcm.setTorqueKernel(tolerance=1e-3)
    ref = FCMOracle(o32, L, cells, tolerance=1e-3, viscosity=visc, seed=seed)
    okt, sup = ref.torque_kernel(tolerance=1e-3)
    assert kt.support[0] == sup
    dp, df, dt_ = (torch.from_numpy(x).cuda() for x in (pos, force, torque))
    for T, pref, f in [(0.0, 0.0, df), (0.6, 2.0, df), (0.0, 0.0, None)]:
        v, w = fcm.computeHydrodynamicDisplacementsTorque(dp, f, dt_, n, T, pref)
        rv, rw = ref.displacements_torque(pos, None if f is None else force, torque, okt, T, pref)
        assert np.linalg.norm(v.cpu().numpy() - rv) <= 1e-5 * np.linalg.norm(rv)
        assert np.linalg.norm(w.cpu().numpy() - rw) <= 1e-5 * np.linalg.norm(rw)
    # without torques the call is the plain one
    v0, _ = fcm.computeHydrodynamicDisplacementsTorque(dp, df, None, n, 0.0, 0.0)
    v1 = fcm.computeHydrodynamicDisplacements(dp, df, n, 0.0, 0.0)
    assert (v0 - v1).abs().max().item() <= 1e-6 * v1.abs().max().item()   # same path; the tile binning order is not fixed


def test_rotational_self_mobility(hip):
    """One particle, torque along each axis, box 64 a: omega = tau/(8 pi eta a^3) up to the periodic-image correction
    ~ (4 pi/3)(a/L)^3 ~ 1e-5 and the discretisation at tolerance 1e-4."""
    cells, visc = [128, 128, 128], 0.9
    k, a = hip.Kernels.Gaussian(1.0, 1e-4)
    L = 128.0
    fcm = hip.BDHI.FCM_impl(hip.Box(L), cells, k, visc, 3, a)
    fcm.setTorqueKernel(tolerance=1e-4)
    pos = torch.tensor([[3.3, -7.1, 11.7, 0.0]], dtype=torch.float32, device="cuda")
    expect = 1.0 / (8 * math.pi * visc * a ** 3)
    for d in range(3):
        tq = torch.zeros((1, 4), dtype=torch.float32, device="cuda")
        tq[0, d] = 1.0
        v, w = fcm.computeHydrodynamicDisplacementsTorque(pos, None, tq, 1, 0.0, 0.0)
        w = w.cpu().numpy()[0]
        assert abs(w[d] / expect - 1) < 1e-2, (d, w, expect)      # measured 4e-3: truncated torque window on the h = 1 grid
        assert np.abs(np.delete(w, d)).max() < 1e-3 * expect
        assert np.abs(v.cpu().numpy()).max() < 1e-3 * expect        # a torque does not translate an isolated sphere


def test_orientation_update_matches_oracle(hip, o32):
    import ctypes as C
    from oracle.oracle import _p
    from uammd_amd._lib import check
    from uammd_amd.md import _ptr, current_stream
    n, dt = 2000, 0.03
    rng = np.random.default_rng(9)
    pos = rng.normal(0, 3, (n, 4)).astype(np.float32)
    q = rng.normal(0, 1, (n, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    v = rng.normal(0, 1, (n, 3)).astype(np.float32)
    w = rng.normal(0, 2, (n, 3)).astype(np.float32)
    w[0] = 0                                                       # the identity-rotation branch
    dpos, dq, dv, dw = (torch.from_numpy(x.copy()).cuda() for x in (pos, q, v, w))
    check(hip.load().uammd_fcm_euler_maruyama_dir(_ptr(dpos), _ptr(dq), None, _ptr(dv), _ptr(dw), n, dt, current_stream()))
    rp, rq = pos.copy(), q.copy()
    o32.lib.oracle_fcm_euler_maruyama_dir(_p(rp), _p(rq), None, _p(v), _p(w), n, C.c_float(dt))
    assert np.array_equal(dpos.cpu().numpy(), rp)
    assert np.abs(dq.cpu().numpy() - rq).max() <= 2e-6
    assert np.abs(np.linalg.norm(dq.cpu().numpy(), axis=1) - 1).max() <= 1e-5   # unit quaternions stay unit


def test_fcm_integrator_with_orientations_matches_oracle(hip, o32):
    """BDHI::FCMIntegrator::forwardTime with dir allocated (BDHI_FCM.cu:7-119): forces and torques are reset, the
    interactors write both, positions and quaternions advance with the linear/angular velocities of the torque path."""
    import ctypes as C
    from oracle.fcm import FCMOracle
    from oracle.oracle import _p
    n, L, cells, dt, visc, seed = 300, 32.0, [32, 32, 32], 0.02, 1.1, 4242
    rng = np.random.default_rng(21)
    pos = np.zeros((n, 4), np.float32)
    pos[:, :3] = rng.uniform(-L / 2, L / 2, (n, 3))
    force = np.zeros((n, 4), np.float32)
    force[:, :3] = rng.normal(0, 1, (n, 3))
    torque = np.zeros((n, 4), np.float32)
    torque[:, :3] = rng.normal(0, 1, (n, 3))
    q0 = rng.normal(0, 1, (n, 4)).astype(np.float32)
    q0 /= np.linalg.norm(q0, axis=1, keepdims=True)

    pd = hip.ParticleData(n)
    pd.setPos(pos)
    pd.getDir("write").copy_(torch.from_numpy(q0))

    class Both(hip.Interactor):
        def __init__(self, pd):
            self.pd = pd

        def sum(self, force=False, energy=False, virial=False):
            # the integrator must have zeroed both before calling the interactors
            assert float(self.pd.getForce("read").abs().max()) == 0.0 and float(self.pd.getTorque("read").abs().max()) == 0.0
            self.pd.getForce("readwrite").add_(torch.from_numpy(globals_["force"]).cuda())
            self.pd.getTorque("readwrite").add_(torch.from_numpy(globals_["torque"]).cuda())

    globals_ = {"force": force, "torque": torque}
    par = hip.BDHI.FCMIntegrator.Parameters(temperature=0.0, viscosity=visc, tolerance=1e-3, dt=dt, box=hip.Box(L), cells=cells,
                                            seed=seed)
    integ = hip.BDHI.FCMIntegrator(pd, par)
    integ.addInteractor(Both(pd))
    integ.forwardTime()
    torch.cuda.synchronize()

    ref = FCMOracle(o32, L, cells, tolerance=1e-3, viscosity=visc, seed=seed)
    okt, _ = ref.torque_kernel(tolerance=1e-3)
    rv, rw = ref.displacements_torque(pos, force, torque, okt, 0.0, 0.0)
    rp, rq = pos.copy(), q0.copy()
    rv32, rw32 = np.ascontiguousarray(rv, np.float32), np.ascontiguousarray(rw, np.float32)
    o32.lib.oracle_fcm_euler_maruyama_dir(_p(rp), _p(rq), None, _p(rv32), _p(rw32), n, C.c_float(dt))
    gp, gq = pd.getPos("read").cpu().numpy(), pd.getDir("read").cpu().numpy()
    assert np.abs(gp - rp).max() <= 1e-5 * dt * np.abs(rv).max() + 2e-6
    assert np.abs(gq - rq).max() <= 1e-5
    assert np.abs(gq - q0).max() > 1e-4   # the orientations did move
