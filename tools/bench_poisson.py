import time, numpy as np, torch, sys
sys.path.insert(0, '.')
import uammd_amd as hip
n, L = 1000000, 128.0
rng = np.random.default_rng(1)
pos = np.zeros((n,4), np.float32); pos[:, :3] = rng.uniform(-0.5,0.5,(n,3))*L
q = rng.normal(0,1,n).astype(np.float32); q -= q.mean()
pd = hip.ParticleData(n); pd.setPos(pos); pd.getCharge("write").copy_(torch.from_numpy(q))
for split in (0.5, 1.0):
    par = hip.Poisson.Parameters(box=hip.Box(L), epsilon=1.0, gw=0.5, tolerance=1e-4, split=split)
    p = hip.Poisson(pd, par)
    print("split", split, "cells", p.cells, "support", p.support, "rc", p.nearFieldCutOff)
    for _ in range(3): p.sum(force=True)
    torch.cuda.synchronize(); t=time.time()
    for _ in range(10): p.sum(force=True)
    torch.cuda.synchronize(); print("ms/sum", (time.time()-t)*100)
