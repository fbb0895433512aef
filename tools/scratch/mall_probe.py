import torch, time
dev = "cuda"
def bw(nbytes_each, reps=50):
    n = nbytes_each // 4
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    for _ in range(5): b.copy_(a)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): b.copy_(a)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    return 2 * nbytes_each / (ms * 1e-3) / 1e12, ms * 1e3
def inplace(nbytes, reps=50):
    n = nbytes // 4
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    for _ in range(5): a.mul_(1.0001)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): a.mul_(1.0001)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    return 2 * nbytes / (ms * 1e-3) / 1e12, ms * 1e3
for mb in (16, 32, 64, 75, 100, 150, 256, 512, 1024):
    t, us = bw(mb << 20)
    t2, us2 = inplace(2 * (mb << 20))
    print("copy %4d MB -> %4d MB: %5.2f TB/s (%6.1f us)   in-place x*=c over %4d MB: %5.2f TB/s (%6.1f us)" % (mb, mb, t, us, 2 * mb, t2, us2))
