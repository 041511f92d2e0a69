import torch, time
x = torch.empty(128*640*640*3, dtype=torch.uint8).pin_memory()
d = torch.empty_like(x, device="cuda")
for n in (1, 4):
    torch.cuda.synchronize(); t=time.time()
    for _ in range(5):
        if n == 1: d.copy_(x, non_blocking=True)
        else:
            for c in range(n):
                k = x.numel()//n
                d[c*k:(c+1)*k].copy_(x[c*k:(c+1)*k], non_blocking=True)
    torch.cuda.synchronize(); dt=(time.time()-t)/5
    print(n, "chunks: H2D GB/s", x.numel()/dt/1e9, "ms", dt*1e3)
