import torch, time
x = torch.empty(256*1024*1024, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
for n in (64, 256):
    a, b = x[: n*1024*1024//4], y[: n*1024*1024//4]
    b.copy_(a); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): b.copy_(a)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"copy {n} MiB: {2 * n * 1.048576e6 / dt / 1e12:.2f} TB/s (read + write), {dt*1e6:.0f} us")
