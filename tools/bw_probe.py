import torch, time
dev = torch.device("cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (256, 1024, 4096):
    nbytes = mb << 20
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    b = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    tf = t(lambda: a.fill_(1.0))
    tc = t(lambda: b.copy_(a))
    ts = t(lambda: a.sum())
    print("%5d MiB: fill %.0f GB/s  copy %.0f GB/s (R+W)  read(sum) %.0f GB/s" % (mb, nbytes / tf / 1e9, 2 * nbytes / tc / 1e9, nbytes / ts / 1e9))
