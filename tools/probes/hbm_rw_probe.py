"""Read, write and copy rates of HBM on this card with torch's own stream kernels (no code of ours): what a pure store stream, a pure load
stream and a copy reach at 64 MB (Infinity-Cache sized) and 2 GB.  The BatchNorm stream kernels and the convolution epilogues are priced
against these, not against the 8 TB/s pin rate.   python tools/probes/hbm_rw_probe.py"""
import torch
d = torch.device('cuda', 0)


def rate(fn, nbytes, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e12


for mb in (42, 64, 256, 2048):
    n = mb * (1 << 20) // 2
    a = torch.empty(n, dtype=torch.float16, device=d).normal_()
    b = torch.empty_like(a)
    w = rate(lambda: b.fill_(1.0), n * 2)
    r = rate(lambda: a.sum(dtype=torch.float32), n * 2)
    c = rate(lambda: b.copy_(a), n * 4)
    s = rate(lambda: torch.add(a, a, out=b), n * 4)
    print('%5d MB: store stream %.2f TB/s   load stream %.2f TB/s   copy %.2f TB/s (read + write bytes)   y = x + x %.2f TB/s' % (mb, w, r, c, s), flush=True)
