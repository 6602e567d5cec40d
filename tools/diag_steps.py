"""Wall time of consecutive chunks of LJ steps (diagnostic): python tools/diag_steps.py [chunk] [nchunks]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import uammd_amd as hip
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 25
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 40
n, L = 1_000_000, 107.7217345
pd, box, pot, verlet, pf, _ = bench.lj_setup(hip, n, L, seed=1234, nl="cell")
verlet.forwardTime()
torch.cuda.synchronize()
out = []
sort_at = int(sys.argv[3]) if len(sys.argv) > 3 else -1
for c in range(nch):
    if c == sort_at:
        t0 = time.perf_counter()
        pd.sortParticles()
        torch.cuda.synchronize()
        print(f"sortParticles before chunk {c}: {(time.perf_counter() - t0) * 1e3:.2f} ms")
    t0 = time.perf_counter()
    for _ in range(chunk):
        verlet.forwardTime()
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / chunk * 1e3)
print(" ".join(f"{x:.3f}" for x in out))
