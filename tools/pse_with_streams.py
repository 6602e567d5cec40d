import sys, os, json, io, contextlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import uammd_amd as hip
import bench
ns = int(sys.argv[1])
keep = [torch.cuda.Stream() for _ in range(ns)]
for s in keep:
    with torch.cuda.stream(s):
        torch.zeros(16, device="cuda").add_(1)
torch.cuda.synchronize()
class A: pass
a = A(); a.pse_steps = 50; a.no_cpu_baseline = True
out = bench.run_pse(hip, a)
print(ns, "extra streams:", out["ms_per_step"])
