"""A/B of the pixel-stationary 1 x 1 kernel (csrc/conv_px.hip, sn_conv_px) against the 160 x 128 tile kernel on the layers it takes
over at the BASELINE C2 batch (20 chips): microseconds per launch between HIP events on the launch stream, operands rotated over
four buffer sets (no launch finds its own previous operands in L2), outputs and BatchNorm partials compared bit for bit.

    python tools/conv_px_ab.py [iters=40]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from sniper_amd import hip

dev = torch.device('cuda:0')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
SETS = 4


def rnd(rs, *shape, scale=1.0):
    return torch.from_numpy((rs.standard_normal(shape) * scale).astype(np.float32)).to(dev).half()


def timed(fn):
    for k in range(SETS):
        fn(k)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for it in range(iters):
        fn(it % SETS)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def fwd_case(N, H, C, O):
    rs = np.random.RandomState(C + O)
    M = N * H * H
    xs = [rnd(rs, N, H, H, C) for _ in range(SETS)]
    w = rnd(rs, O, 1, C, scale=1.0 / np.sqrt(C))
    geom = (N, H, H, C, C, O, O, 0, 1, 1, 1, 0, 1)
    nblk = hip.query('sn_conv_fwd_stats_blocks', *geom)
    res = {}
    for on in (0, 1, 2):
        hip.call('sn_conv_px', on)
        ys = [torch.empty((N, H, H, O), dtype=torch.float16, device=dev) for _ in range(SETS)]
        ps = [torch.zeros((nblk, 2, O), dtype=torch.float32, device=dev) for _ in range(SETS)]
        us = timed(lambda k: hip.call('sn_conv_fwd_stats', xs[k], w, None, None, ys[k], *geom, 0, ps[k], hip.stream()))
        res[on] = (us, ys, ps)
    same = all(torch.equal(a, b) and torch.equal(a, c) for a, b, c in zip(res[0][1], res[1][1], res[2][1])) and \
        all(torch.equal(a, b) and torch.equal(a, c) for a, b, c in zip(res[0][2], res[1][2], res[2][2]))
    flop = 2.0 * M * C * O
    mb = (M * C + M * O + O * C) * 2 / 1e6
    print('sn_conv_fwd_stats N%d %dx%d C%d->%d: tile %.1f us, px %.1f us (%.2fx), staggered %.1f us (%.2fx)  %.0f -> %.0f TFLOP/s, %.1f MB: %.2f -> %.2f TB/s  bit-equal %s'
          % (N, H, H, C, O, res[0][0], res[1][0], res[0][0] / res[1][0], res[2][0], res[0][0] / res[2][0], flop / res[0][0] / 1e6,
             flop / res[2][0] / 1e6, mb, mb / res[0][0], mb / res[2][0], same), flush=True)


def dgrad_case(N, H, C, O):
    """dx (C channels) from dy (O channels, the contraction), BatchNorm-backward reduction over bn_x fused"""
    rs = np.random.RandomState(C + O + 1)
    M = N * H * H
    dys = [rnd(rs, N, H, H, O) for _ in range(SETS)]
    bnxs = [rnd(rs, N, H, H, C) for _ in range(SETS)]
    wt = rnd(rs, C, 1, O, scale=1.0 / np.sqrt(O))
    f = lambda a: torch.from_numpy(a.astype(np.float32)).to(dev)
    scale, shift, mean = f(rs.uniform(0.5, 1.5, C)), f(rs.uniform(-0.5, 2.5, C)), f(rs.standard_normal(C) * 0.1)
    geom = (N, H, H, C, C, O, O, 0, 1, 1, 1, 0, 1)
    nblk = hip.query('sn_conv_dgrad_bn_blocks', *geom)
    res = {}
    for on in (0, 1, 2):
        hip.call('sn_conv_px', on)
        dxs = [torch.empty((N, H, H, C), dtype=torch.float16, device=dev) for _ in range(SETS)]
        ps = [torch.zeros((nblk, 2, C), dtype=torch.float32, device=dev) for _ in range(SETS)]
        us = timed(lambda k: hip.call('sn_conv_dgrad_bn', dys[k], wt, None, dxs[k], *geom, bnxs[k], C, scale, shift, mean, 1, ps[k], hip.stream()))
        res[on] = (us, dxs, ps)
    same = all(torch.equal(a, b) and torch.equal(a, c) for a, b, c in zip(res[0][1], res[1][1], res[2][1])) and \
        all(torch.equal(a, b) and torch.equal(a, c) for a, b, c in zip(res[0][2], res[1][2], res[2][2]))
    flop = 2.0 * M * C * O
    mb = (M * O + 2 * M * C + O * C) * 2 / 1e6
    print('sn_conv_dgrad_bn N%d %dx%d dx C%d <- dy C%d: tile %.1f us, px %.1f us (%.2fx), staggered %.1f us (%.2fx)  %.0f -> %.0f TFLOP/s, %.1f MB: %.2f -> %.2f TB/s  bit-equal %s'
          % (N, H, H, C, O, res[0][0], res[1][0], res[0][0] / res[1][0], res[2][0], res[0][0] / res[2][0], flop / res[0][0] / 1e6,
             flop / res[2][0] / 1e6, mb, mb / res[0][0], mb / res[2][0], same), flush=True)


if __name__ == '__main__':
    fwd_case(20, 32, 256, 1024)
    dgrad_case(20, 32, 1024, 256)
    fwd_case(20, 64, 128, 512)
    dgrad_case(20, 64, 512, 128)
    fwd_case(20, 32, 256, 512)
    hip.call('sn_conv_px', int(os.environ.get('SNIPER_CONV_PX', '0') or 0))
