import torch, time
for gb in (1, 4, 10.9, 36):
    n = int(gb * 1e9 / 4)
    x = torch.empty(n, dtype=torch.float32, device="cuda:0")
    for _ in range(2): x.fill_(0.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): x.fill_(1.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"fill {gb} GB: {dt*1e3:.3f} ms  {gb/dt/1e3:.2f} TB/s")
    del x
