import torch, time
x = torch.empty(210 * 1024 * 1024, dtype=torch.uint8, device="cuda")
y = torch.empty_like(x)
for name, fn in (("fill", lambda: x.fill_(1)), ("copy", lambda: y.copy_(x))):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(name, "210 MiB:", round(dt * 1e6, 1), "us ->", round(x.numel() * (2 if name == "copy" else 1) / dt / 1e12, 2), "TB/s")
