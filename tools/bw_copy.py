import torch
for mb in (25.6, 51.2, 205, 820):
    n = int(mb * 1e6 / 4)
    a = torch.zeros(n, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
    for _ in range(5): b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 50 * 1e-3
    print(f"{mb} MB copy: {t*1e6:.1f} us  read+write {2*mb*1e6/t/1e12:.2f} TB/s")
