"""Sweep the convolution kernel configurations (sn_conv_tune) over the R101 / batch-20 layer shapes (BASELINE C2):
for every (layer shape, direction) time each configuration with HIP events, check its output against the
register-staged kernel (cfg 0) and print the table conv_plan()'s built-in choice is filled from.

    python tools/conv_tune.py [--batch 20] [--iters 20] [--cfgs 0,14,16,18] [--only stage3]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sniper_amd import hip  # noqa: E402

# name, H, W, Cin, Cout, K, stride, pad, dil, fwd launches / step, dgrad launches / step   (R101 C4 trunk + RPN + heads)
LAYERS = [
    ('s1 1x1 64->256 @128', 128, 128, 64, 256, 1, 1, 0, 1, 3, 0),
    ('s1 3x3 64->64 @128', 128, 128, 64, 64, 3, 1, 1, 1, 3, 0),
    ('s1 1x1 256->64 @128', 128, 128, 256, 64, 1, 1, 0, 1, 2, 0),
    ('s2u1 1x1 256->128 @128', 128, 128, 256, 128, 1, 1, 0, 1, 1, 0),
    ('s2u1 3x3s2 128->128', 128, 128, 128, 128, 3, 2, 1, 1, 1, 1),
    ('s2u1 sc 1x1s2 256->512', 128, 128, 256, 512, 1, 2, 0, 1, 1, 0),
    ('s2 1x1 512->128 @64', 64, 64, 512, 128, 1, 1, 0, 1, 3, 3),
    ('s2 3x3 128->128 @64', 64, 64, 128, 128, 3, 1, 1, 1, 3, 3),
    ('s2 1x1 128->512 @64', 64, 64, 128, 512, 1, 1, 0, 1, 4, 4),
    ('s3u1 1x1 512->256 @64', 64, 64, 512, 256, 1, 1, 0, 1, 1, 1),
    ('s3u1 3x3s2 256->256', 64, 64, 256, 256, 3, 2, 1, 1, 1, 1),
    ('s3u1 sc 1x1s2 512->1024', 64, 64, 512, 1024, 1, 2, 0, 1, 1, 1),
    ('s3 1x1 1024->256 @32', 32, 32, 1024, 256, 1, 1, 0, 1, 22, 22),
    ('s3 3x3 256->256 @32', 32, 32, 256, 256, 3, 1, 1, 1, 22, 22),
    ('s3 1x1 256->1024 @32', 32, 32, 256, 1024, 1, 1, 0, 1, 23, 23),
    ('s4u1 1x1 1024->512 @32', 32, 32, 1024, 512, 1, 1, 0, 1, 1, 1),
    ('s4 off 3x3d2 512->72', 32, 32, 512, 72, 3, 1, 2, 2, 3, 3),
    ('s4 deform gemm 4608->512', 32, 32, 4608, 512, 1, 1, 0, 1, 3, 3),
    ('s4 1x1 512->2048 @32', 32, 32, 512, 2048, 1, 1, 0, 1, 3, 3),
    ('s4u1 sc 1x1 1024->2048', 32, 32, 1024, 2048, 1, 1, 0, 1, 1, 1),
    ('s4 1x1 2048->512 @32', 32, 32, 2048, 512, 1, 1, 0, 1, 2, 2),
    ('rpn 3x3 3072->512 @32', 32, 32, 3072, 512, 3, 1, 1, 1, 1, 1),
    ('rpn bbox 1x1 512->84', 32, 32, 512, 84, 1, 1, 0, 1, 1, 1),
    ('conv_new_1 1x1 2048->256', 32, 32, 2048, 256, 1, 1, 0, 1, 1, 1),
    ('fc_new_1 12544->1024 x6000', 0, 0, 12544, 1024, 1, 1, 0, 1, 1, 1),
    ('fc_new_2 1024->1024 x6000', 0, 0, 1024, 1024, 1, 1, 0, 1, 1, 1),
    ('fc offset 12544->98 x6000', 0, 0, 12544, 98, 1, 1, 0, 1, 1, 1),
    ('fc cls 1024->81 x6000', 0, 0, 1024, 81, 1, 1, 0, 1, 1, 1),
]


def timeit(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=20)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--cfgs', default='0,4,5,6,7,14,16,18')
    ap.add_argument('--only', default='')
    ap.add_argument('--cold', type=int, default=0, help='MB of distinct operand sets to cycle through (> L2 + Infinity Cache = 288): '
                    'every launch then reads operands no XCD has cached, as in the training step, where the producer of a tensor '
                    'ran on other XCDs and 30 ms of other traffic separate two uses of a weight')
    ap.add_argument('--insitu', action='store_true', help='call the layers the way the training step does: forward with the '
                    'BatchNorm-statistics epilogue (sn_conv_fwd_stats; + residual on the 1x1 expansions), data gradient '
                    'accumulating into the trunk gradient where Nout = 4 K')
    a = ap.parse_args()
    cfgs = [int(c) for c in a.cfgs.split(',')]
    d = torch.device('cuda', 0)
    B = a.batch
    g = torch.Generator(device=d)
    g.manual_seed(0)
    h = lambda *s: (torch.randn(*s, device=d, generator=g) * 0.5).half()
    print('device', torch.cuda.get_device_name(0), 'batch', B, 'cfgs', cfgs, flush=True)
    tot = {c: 0.0 for c in cfgs}
    best_tot = 0.0
    table = []
    for (name, H, W, C, O, K, s, p, dl, nf, nd) in LAYERS:
        if a.only and a.only not in name:
            continue
        if H == 0:
            N, H, W = B * 300, 1, 1
        else:
            N = B
        Ho, Wo = (H + 2 * p - dl * (K - 1) - 1) // s + 1, (W + 2 * p - dl * (K - 1) - 1) // s + 1
        M = N * Ho * Wo
        fl = 2.0 * M * O * C * K * K
        Op = (O + 7) // 8 * 8
        per_set = 2 * (N * H * W * C + O * K * K * C + N * Ho * Wo * O)
        nbuf = max(1, min(64, -(-a.cold * (1 << 20) // per_set))) if a.cold else 1
        xs, ws_ = [h(N, H, W, C) for _ in range(nbuf)], [h(O, K * K, C) for _ in range(nbuf)]
        ys = [torch.empty((N, Ho, Wo, O), dtype=torch.float16, device=d) for _ in range(nbuf)]
        dys, wts = [h(N, Ho, Wo, Op) for _ in range(nbuf)], [h(C, K * K, Op) for _ in range(nbuf)]
        dxs = [torch.empty_like(xs[0]) for _ in range(nbuf)]
        x, w, y, dy, wt, dx = xs[0], ws_[0], ys[0], dys[0], wts[0], dxs[0]
        ctr = [0]
        for direction, cnt in (('fwd', nf), ('dgrad', nd)):
            if cnt == 0:
                continue
            expand = K == 1 and s == 1 and O == 4 * C          # conv3 of a bottleneck: residual add in the epilogue
            reduce_ = K == 1 and s == 1 and C == 4 * O         # conv1: its data gradient accumulates into the trunk gradient
            if direction == 'fwd':
                part = torch.empty((4096, 2, O), dtype=torch.float32, device=d)
                ress = [h(N, Ho, Wo, O) for _ in range(nbuf)] if (a.insitu and expand) else None

                def run():
                    i = ctr[0] % nbuf
                    ctr[0] += 1
                    if a.insitu and hip.query('sn_conv_fwd_stats_blocks', N, H, W, C, C, O, O, O if ress else 0, K, K, s, p, dl) > 0:
                        hip.call('sn_conv_fwd_stats', xs[i], ws_[i], None, ress[i] if ress else None, ys[i], N, H, W, C, C, O, O,
                                 O if ress else 0, K, K, s, p, dl, 0, part, hip.stream())
                    else:
                        hip.call('sn_conv_fwd', xs[i], ws_[i], None, None, ys[i], N, H, W, C, C, O, O, O, K, K, s, p, dl, 0, 0, hip.stream())
                out = y
            else:
                def run():
                    i = ctr[0] % nbuf
                    ctr[0] += 1
                    acc = dxs[(i + 1) % nbuf] if (a.insitu and reduce_) else None
                    hip.call('sn_conv_dgrad', dys[i], wts[i], acc, dxs[i], N, H, W, C, C, Op, Op, C, K, K, s, p, dl, 0, hip.stream())
                out = dx
            ref = None
            row = {}
            for c in cfgs:
                hip.call('sn_conv_tune', c)
                out.zero_()
                ctr[0] = 0
                try:
                    run()
                    torch.cuda.synchronize()
                except Exception as e:   # noqa: BLE001
                    print('  cfg %d failed: %s' % (c, e), flush=True)
                    continue
                o = out.float()
                if ref is None:
                    ref = o.clone()
                    err = 0.0
                else:
                    # (in-situ mode accumulates into buffers earlier runs wrote: outputs are compared in the plain mode only)
                    err = 0.0 if a.insitu else float((o - ref).abs().max() / ref.abs().max().clamp_min(1e-6))
                us = timeit(run, max(a.iters, 2 * nbuf) if a.cold else a.iters)
                row[c] = (us, err)
            hip.call('sn_conv_tune', -1)
            bc = min(row, key=lambda c: row[c][0])
            cells = ' '.join('%d:%6.1f%s' % (c, row[c][0], '' if row[c][1] < 2e-3 else '!ERR%.1e' % row[c][1]) for c in row)
            print('%-28s %-5s M=%6d N=%4d K=%5d x%2d | %s | best %d %.1f us %.0f TF/s (legacy %.0f TF/s)' % (
                name, direction, M, Op if direction == 'dgrad' else O, C * K * K, cnt, cells, bc, row[bc][0], fl / row[bc][0] / 1e6,
                fl / row[cfgs[0]][0] / 1e6), flush=True)
            for c in row:
                tot[c] += row[c][0] * cnt
            best_tot += row[bc][0] * cnt
            table.append((name, direction, M, O if direction == 'fwd' else C, (Op if direction == 'dgrad' else C) * K * K // 64, bc))
    print('per-step totals (ms): ' + ' '.join('%d:%.2f' % (c, tot[c] / 1e3) for c in cfgs) + ' | best-per-layer %.2f' % (best_tot / 1e3))
    print('table (direction, M, Nout, nk) -> cfg:')
    for t in table:
        print('   ', t)


if __name__ == '__main__':
    main()
