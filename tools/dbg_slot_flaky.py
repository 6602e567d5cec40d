"""repro loop for tests/test_gpu_ibm_fcm.py::test_fcm_step_slot_layout[cells2-100000-0]: which step, which particles"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import uammd_amd as hip

cells, n = (128, 128, 128), 100000
L = np.asarray(cells, np.float32)
k, a_eff = hip.Kernels.Gaussian(1.0, 1e-3)
dt = 0.01
rng = np.random.default_rng(n)
pos = np.zeros((n, 4), np.float32)
pos[:, :3] = rng.uniform(-0.5, 0.5, (n, 3)) * L
force = np.zeros((n, 4), np.float32)
force[:, :3] = rng.normal(0, 1, (n, 3))

def run(slots, T, steps=14):
    fcm = hip.BDHI.FCM_impl(hip.Box(L), cells, k, 0.9, 5, a_eff)
    fcm.set_option("slots", 1 if slots else 0)
    fcm.set_option("slot_refresh", 4)
    dp, df = torch.from_numpy(pos.copy()).cuda(), torch.from_numpy(force).cuda()
    v = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    traj, ps = [], []
    for s in range(steps):
        ff = None if s == 9 else df
        ps.append(dp.cpu().numpy().copy())
        fcm.stepEulerMaruyama(dp, ff, n, T, 1 / math.sqrt(dt), dt, out=v, positions_kept=s > 0)
        traj.append(v.cpu().numpy().copy())
    torch.cuda.synchronize()
    return traj, ps

for rep in range(int(os.environ.get("REPS", 8))):
    for T in (0.0, 0.6):
        va, pa = run(True, T)
        vb, pb = run(False, T)
        scale = max(np.abs(x).max() for x in vb)
        for s, (x, y) in enumerate(zip(va, vb)):
            d = np.abs(x - y).max(axis=1)
            if d.max() > 2e-5 * scale:
                bad = np.nonzero(d > 2e-5 * scale)[0]
                print(f"rep {rep} T {T} step {s}: max {d.max():.3e} scale {scale:.2f}; {bad.size} particles over the bar; first {bad[:8]}")
                for i in bad[:4]:
                    cell = np.floor(pa[s][i, :3] + L / 2)
                    print("   particle", i, "pos", pa[s][i, :3], "cell", cell, "cell mod 8", cell % 8, "dv", x[i] - y[i])
                break
        else:
            continue
        break
print("done")
