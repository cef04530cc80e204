"""Why is a stage-3 convolution slower inside the training step than in tools/conv_tune.py?  Times the three stage-3 layer
shapes (batch 20) the way the step calls them: plain forward, forward + BatchNorm-statistics epilogue, + residual, on operands
cycled through > 288 MB (cold L2 / Infinity Cache), and interleaved with the BatchNorm apply pass that precedes each one."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sniper_amd import hip  # noqa: E402

d = torch.device('cuda', 0)
B, H = 20, 32
h = lambda *s: (torch.randn(*s, device=d) * 0.5).half()


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (name, C, O, K, p) in (('1x1 1024->256', 1024, 256, 1, 0), ('3x3 256->256', 256, 256, 3, 1), ('1x1 256->1024', 256, 1024, 1, 0)):
    nb = 24
    xs, ws, ys, rs = [h(B, H, H, C) for _ in range(nb)], [h(O, K * K, C) for _ in range(nb)], [h(B, H, H, O) for _ in range(nb)], [h(B, H, H, O) for _ in range(nb)]
    nblk = hip.query('sn_conv_fwd_stats_blocks', B, H, H, C, C, O, O, O, K, K, 1, p, 1)
    part = torch.empty((max(nblk, 1), 2, O), dtype=torch.float32, device=d)
    sc, sh = torch.ones(C, device=d), torch.zeros(C, device=d)
    M = B * H * H
    c = [0]

    def plain():
        i = c[0] % nb; c[0] += 1
        hip.call('sn_conv_fwd', xs[i], ws[i], None, None, ys[i], B, H, H, C, C, O, O, O, K, K, 1, p, 1, 0, 0, hip.stream())

    def stats():
        i = c[0] % nb; c[0] += 1
        hip.call('sn_conv_fwd_stats', xs[i], ws[i], None, None, ys[i], B, H, H, C, C, O, O, 0, K, K, 1, p, 1, 0, part, hip.stream())

    def stats_res():
        i = c[0] % nb; c[0] += 1
        hip.call('sn_conv_fwd_stats', xs[i], ws[i], None, rs[i], ys[i], B, H, H, C, C, O, O, O, K, K, 1, p, 1, 0, part, hip.stream())

    def bn_then_conv():
        i = c[0] % nb; c[0] += 1
        j = (i + 1) % nb
        hip.call('sn_bn_apply', xs[j], xs[i], M, C, C, C, sc, sh, 1, hip.stream())
        hip.call('sn_conv_fwd_stats', xs[i], ws[i], None, None, ys[i], B, H, H, C, C, O, O, 0, K, K, 1, p, 1, 0, part, hip.stream())

    def bn_only():
        i = c[0] % nb; c[0] += 1
        j = (i + 1) % nb
        hip.call('sn_bn_apply', xs[j], xs[i], M, C, C, C, sc, sh, 1, hip.stream())

    t = {k: timeit(f, 48) for k, f in (('plain', plain), ('stats', stats), ('stats+res', stats_res), ('bn_apply', bn_only), ('bn_apply+conv', bn_then_conv))}
    print('%-16s plain %6.1f  stats %6.1f  stats+res %6.1f | bn_apply %6.1f  bn_apply -> conv(stats) %6.1f (sum of parts %6.1f) us' % (
        name, t['plain'], t['stats'], t['stats+res'], t['bn_apply'], t['bn_apply+conv'], t['bn_apply'] + t['stats']), flush=True)
