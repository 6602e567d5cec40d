"""CPU tests: Saru stream and integrator restatements.  The reference ships no Saru vector (parity of the raw
stream is pinned by the committed golden file generated at round 1 + the statistical checks here); the integrators are
checked against closed forms."""
import math
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_saru_golden(o32):
    g = np.load(os.path.join(GOLD, "saru_u32.npz"))
    for row, s in zip(g["three"], g["seeds"]):
        assert np.array_equal(o32.saru_u32([int(x) for x in s], 16), row)
    for row, s in zip(g["one"], g["seeds"]):
        assert np.array_equal(o32.saru_u32([int(s[0])], 16), row)
    for row, s in zip(g["two"], g["seeds"]):
        assert np.array_equal(o32.saru_u32([int(s[0]), int(s[1])], 16), row)
    assert np.array_equal(o32.saru_f_range(1234, -0.5, 0.5, 60), g["f_range_1234"])


def test_saru_python_reimplementation(o32):
    """An independent Python transcription of the 3-seed constructor + u32 (saruprng.cuh:253-273, :340-347)."""
    M = 0xFFFFFFFF

    def sar(x, k):
        x &= M
        return ((x - (1 << 32)) >> k) & M if x & 0x80000000 else x >> k

    def saru3(s1, s2, s3, n):
        s3 ^= ((s1 << 7) & M) ^ (s2 >> 6)
        s2 = (s2 + ((s1 >> 4) ^ (s3 >> 15))) & M
        s1 ^= (((s2 << 9) & M) + ((s3 << 8) & M)) & M
        s3 ^= (0xA5366B4D * ((s2 >> 11) ^ ((s1 << 1) & M))) & M
        s2 = (s2 + 0x72BE1579 * (((s1 << 4) & M) ^ (s3 >> 16))) & M
        s1 ^= (0x3F38A6ED * ((s3 >> 5) ^ sar(s2, 22))) & M
        s2 = (s2 + s1 * s3) & M
        s1 = (s1 + (s3 ^ (s2 >> 2))) & M
        s2 ^= sar(s2, 17)
        state = (0x79dedea3 * (s1 ^ sar(s1, 14))) & M
        w = ((state + s2) & M) ^ sar(state, 8)
        state = (state + w * (w ^ 0xdddf97f5)) & M
        w = (0xABCB96F7 + (w >> 1)) & M
        out = []
        for _ in range(n):
            state = (0x4beb5d59 * state + 0x2600e1f7) & M
            w = (w + 0x8009d14b + (sar(w, 31) & 0xda879add)) & M
            v = ((state ^ (state >> 26)) + w) & M
            out.append(((v ^ (v >> 20)) * 0x6957f5a7) & M)
        return out
    for s in [(0, 0, 0), (1, 2, 3), (0xFFFFFFFF, 5, 0x80000000), (123456789, 987654321, 42)]:
        assert o32.saru_u32(list(s), 8).tolist() == saru3(*s, 8)


def test_saru_statistics(o32):
    u = o32.saru_f_range(99, -1.0, 1.0, 200000)
    assert abs(u.mean()) < 5e-3 and abs(u.var() - 1 / 3) < 5e-3 and u.min() >= -1 and u.max() <= 1
    g = o32.saru_gf((5, 6, 7), 0.0, 2.0, 100000)
    assert abs(g.mean()) < 2e-2 and abs(g.std() - 2.0) < 2e-2


def test_bd_free_diffusion(o32):
    """BD::EulerMaruyama, config C1 shape: <dx^2> = 2 T M dt per step and coordinate; deterministic drift M F dt."""
    n, dt, T, M = 20000, 0.1, 1.0, 1.0 / (6 * math.pi)
    pos = np.zeros((n, 4), np.float32)
    force = np.zeros((n, 4), np.float32)
    force[:, 0] = 2.0
    o32.bd_euler_maruyama(pos, force, M, dt, T, 1, 1234)
    d = pos[:, :3].astype(np.float64)
    assert abs(d[:, 0].mean() - M * 2.0 * dt) < 3e-3
    assert np.all(np.abs(d.var(0) - 2 * T * M * dt) < 0.03 * 2 * T * M * dt)
    p2 = np.zeros((4, 4), np.float32)
    o32.bd_euler_maruyama(p2, force[:4], M, dt, 0.0, 1, 1)
    assert np.allclose(p2[:, 0], M * 2.0 * dt) and np.all(p2[:, 1:] == 0)


def test_gronbech_jensen_deterministic_limit(o32):
    """T = 0, friction = 0: the two GJ half steps are plain velocity Verlet."""
    n, dt = 64, 0.01
    rng = np.random.default_rng(1)
    pos = np.zeros((n, 4), np.float32); pos[:, :3] = rng.normal(0, 1, (n, 3))
    vel = rng.normal(0, 1, (n, 3)).astype(np.float32)
    f0 = np.zeros((n, 4), np.float32); f0[:, :3] = rng.normal(0, 1, (n, 3))
    p, v, f = pos.copy(), vel.copy(), f0.copy()
    o32.verletnvt_gj(1, p, v, f, dt, 0.0, 0.0, 1, 7)
    assert np.all(f == 0)
    assert np.allclose(p[:, :3], pos[:, :3] + dt * vel + 0.5 * dt * dt * f0[:, :3], atol=1e-6)
    f1 = np.zeros((n, 4), np.float32); f1[:, :3] = rng.normal(0, 1, (n, 3))
    f[:] = f1
    o32.verletnvt_gj(2, p, v, f, dt, 0.0, 0.0, 1, 7)
    assert np.allclose(v, vel + 0.5 * dt * (f0[:, :3] + f1[:, :3]), atol=1e-6)


def test_initial_velocities_amplitude(o32):
    v = o32.verletnvt_initial_velocities(50000, math.sqrt(3 * 2.0), 77)
    assert abs(v.std() - math.sqrt(6.0)) < 0.03 and abs(v.mean()) < 0.03


def _bd_free_run(o32, scheme, n, steps, dt, T, M, seed):
    """free particles under one of the three other BD schemes (Integrator/BrownianDynamics.cu:160-387), as their forwardTime strings the calls"""
    pos = np.zeros((n, 4), np.float32)
    force = np.zeros((n, 4), np.float32)
    aux = np.zeros((n, 4), np.float32)
    traj = [pos[:, :3].astype(np.float64).copy()]
    for step in range(1, steps + 1):
        if scheme == "MidPoint":
            o32.bd_midpoint(0, pos, aux, force, M, dt, T, step, seed)
            o32.bd_midpoint(1, pos, aux, force, M, dt, T, step, seed)
        elif scheme == "AdamsBashforth":
            o32.bd_adams_bashforth(pos, aux, force, M, dt, T, step, seed)
        else:
            o32.bd_leimkuhler(pos, force, M, dt, T, step, seed)
        traj.append(pos[:, :3].astype(np.float64).copy())
    return np.array(traj)


@pytest.mark.parametrize("scheme", ["MidPoint", "AdamsBashforth", "Leimkuhler"])
def test_bd_other_schemes_free_diffusion(o32, scheme):
    """the criterion of the reference's test/BD/test.bash: the mean square displacement of free particles grows as 2 D0 t per coordinate
    (D0 = T M).  Leimkuhler's noise (dW_n + dW_(n-1)) / 2 is correlated over one step: <dx^2>(n steps) = (2 n - 1) T M dt, the same slope."""
    n, dt, T, M = 20000, 0.1, 1.0, 1.0 / (6 * math.pi)
    traj = _bd_free_run(o32, scheme, n, 12, dt, T, M, 4321)
    msd = ((traj - traj[0]) ** 2).mean(1)                      # [step][coordinate]
    slope = (msd[12] - msd[4]) / 8
    assert np.all(np.abs(slope / (2 * T * M * dt) - 1) < 0.04), slope / (2 * T * M * dt)
    if scheme == "Leimkuhler":
        assert np.all(np.abs(msd[1] / (T * M * dt) - 1) < 0.04)
    else:
        assert np.all(np.abs(msd[1] / (2 * T * M * dt) - 1) < 0.04)


def test_bd_other_schemes_deterministic_limits(o32):
    """T = 0: with a constant force MidPoint's two sub-steps land where one Euler step does; AdamsBashforth with F_(n-1) = F_n is Euler;
    with F_(n-1) = 0 it moves by 3/2 of it; Leimkuhler without noise is Euler.  Members of a group only, shear included."""
    n, dt, M = 64, 0.05, 0.7
    rng = np.random.default_rng(5)
    pos0 = np.zeros((n, 4), np.float32); pos0[:, :3] = rng.normal(0, 1, (n, 3)); pos0[:, 3] = np.arange(n)
    force = np.zeros((n, 4), np.float32); force[:, :3] = rng.normal(0, 1, (n, 3))
    index = np.arange(0, n, 2, dtype=np.int32)
    euler = pos0.copy()
    o32.bd_euler_maruyama(euler, force, M, dt, 0.0, 1, 1, index=None)
    noshear = euler.copy()
    noshear[1::2] = pos0[1::2]
    mid, aux = pos0.copy(), np.zeros((len(index), 4), np.float32)
    o32.bd_midpoint(0, mid, aux, force, M, dt, 0.0, 1, 1, index=index)
    assert np.array_equal(aux, pos0[index])
    half = pos0[index, :3] + 0.5 * dt * M * force[index, :3]
    assert np.allclose(mid[index, :3], half, atol=1e-6)
    o32.bd_midpoint(1, mid, aux, force, M, dt, 0.0, 1, 1, index=index)
    assert np.allclose(mid[:, :3], noshear[:, :3], atol=1e-6) and np.array_equal(mid[:, 3], pos0[:, 3])
    ab = pos0.copy()
    o32.bd_adams_bashforth(ab, force[index].copy(), force, M, dt, 0.0, 1, 1, index=index)
    assert np.allclose(ab[:, :3], noshear[:, :3], atol=1e-6)
    ab = pos0.copy()
    o32.bd_adams_bashforth(ab, np.zeros((len(index), 4), np.float32), force, M, dt, 0.0, 1, 1, index=index)
    assert np.allclose(ab[index, :3], pos0[index, :3] + 1.5 * dt * M * force[index, :3], atol=1e-6) and np.array_equal(ab[1::2], pos0[1::2])
    lk = pos0.copy()
    o32.bd_leimkuhler(lk, force, M, dt, 0.0, 1, 1, index=index)
    assert np.allclose(lk[:, :3], noshear[:, :3], atol=1e-6)
    K = np.array([[0, 0.3, 0], [0, 0, 0], [0.1, 0, 0]], np.float32)
    sh = pos0.copy()
    o32.bd_leimkuhler(sh, force, M, dt, 0.0, 1, 1, K=K)
    expect = pos0[:, :3] + dt * (pos0[:, :3] @ K.T + M * force[:, :3])
    assert np.allclose(sh[:, :3], expect, atol=1e-6)


def test_bd_leimkuhler_shares_draws_between_steps(o32):
    """this step's second draw is the next step's first (keyed (particle, step, seed)): x_2 - x_0 = B (dW_2 + 2 dW_1 + dW_0)"""
    n, dt, T, M = 50000, 0.2, 1.0, 0.3
    pos = np.zeros((n, 4), np.float32)
    force = np.zeros((n, 4), np.float32)
    o32.bd_leimkuhler(pos, force, M, dt, T, 1, 9)
    o32.bd_leimkuhler(pos, force, M, dt, T, 2, 9)
    var = pos[:, :3].astype(np.float64).var(0)
    assert np.all(np.abs(var / (0.5 * T * M * dt * 6) - 1) < 0.03), var   # 1 + 4 + 1
