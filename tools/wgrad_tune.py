"""Per-layer launches of the weight gradient (sn_conv_wgrad) over the R101 / batch-20 layer shapes (BASELINE C2), modes =
(impl, job_steps) of sn_conv_wgrad_impl: the wave-specialised kernel with its built-in or a forced job length, or the gather
fallback (impl 0); each checked against an fp32 torch contraction on a sub-sampled set of output elements and against the first
mode's result.  (The training step launches TABLES of layers: tools/wgrad_batch_bench.py.)

    python tools/wgrad_tune.py [--batch 20] [--iters 20] [--only s3]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sniper_amd import hip  # noqa: E402

# name, H, W, Cin, Cout, K, stride, pad, dil, wgrad launches / step
LAYERS = [
    ('s2u1 3x3s2 128->128', 128, 128, 128, 128, 3, 2, 1, 1, 1),
    ('s2 1x1 512->128 @64', 64, 64, 512, 128, 1, 1, 0, 1, 3),
    ('s2 3x3 128->128 @64', 64, 64, 128, 128, 3, 1, 1, 1, 3),
    ('s2 1x1 128->512 @64', 64, 64, 128, 512, 1, 1, 0, 1, 4),
    ('s3u1 1x1 512->256 @64', 64, 64, 512, 256, 1, 1, 0, 1, 1),
    ('s3u1 3x3s2 256->256', 64, 64, 256, 256, 3, 2, 1, 1, 1),
    ('s3 1x1 1024->256 @32', 32, 32, 1024, 256, 1, 1, 0, 1, 22),
    ('s3 3x3 256->256 @32', 32, 32, 256, 256, 3, 1, 1, 1, 22),
    ('s3 1x1 256->1024 @32', 32, 32, 256, 1024, 1, 1, 0, 1, 23),
    ('s4u1 1x1 1024->512 @32', 32, 32, 1024, 512, 1, 1, 0, 1, 1),
    ('s4 off 3x3d2 512->72', 32, 32, 512, 72, 3, 1, 2, 2, 3),
    ('s4 deform gemm 4608->512', 32, 32, 4608, 512, 1, 1, 0, 1, 3),
    ('s4 1x1 512->2048 @32', 32, 32, 512, 2048, 1, 1, 0, 1, 3),
    ('s4u1 sc 1x1 1024->2048', 32, 32, 1024, 2048, 1, 1, 0, 1, 1),
    ('s4 1x1 2048->512 @32', 32, 32, 2048, 512, 1, 1, 0, 1, 2),
    ('rpn 3x3 3072->512 @32', 32, 32, 3072, 512, 3, 1, 1, 1, 1),
    ('rpn bbox 1x1 512->84', 32, 32, 512, 84, 1, 1, 0, 1, 1),
    ('conv_new_1 1x1 2048->256', 32, 32, 2048, 256, 1, 1, 0, 1, 1),
    ('fc_new_1 12544->1024 x6000', 0, 0, 12544, 1024, 1, 1, 0, 1, 1),
    ('fc_new_2 1024->1024 x6000', 0, 0, 1024, 1024, 1, 1, 0, 1, 1),
    ('fc cls 1024->81 x6000', 0, 0, 1024, 81, 1, 1, 0, 1, 1),
]
MODES = [(1, 0), (1, 40), (1, 10), (0, 0)]


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


def reference_samples(x, dy, K, s, p, dl, Ho, Wo, cos, cis):
    """fp32 dW[co][tap][ci] for the sampled (co, ci) on the fp16-rounded operands (torch on the device, no kernels of ours)."""
    N, H, W, C = x.shape
    xs = x[..., cis].float()
    ds = dy[..., cos].float()                                  # (N, Ho, Wo, nco)
    out = torch.zeros((len(cos), K * K, len(cis)), device=x.device)
    xp = torch.nn.functional.pad(xs, (0, 0, p, p, p, p))
    for kh in range(K):
        for kw in range(K):
            v = xp[:, kh * dl: kh * dl + (Ho - 1) * s + 1: s, kw * dl: kw * dl + (Wo - 1) * s + 1: s, :]
            out[:, kh * K + kw, :] = torch.einsum('nhwo,nhwi->oi', ds, v)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=20)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--only', default='')
    ap.add_argument('--cold', type=int, default=0, help='MB of distinct operand sets to cycle through (see tools/conv_tune.py)')
    ap.add_argument('--modes', default='', help='e.g. "1/0,1/40,0/0" (impl / job K-steps)')
    a = ap.parse_args()
    global MODES
    if a.modes:
        MODES = [tuple(int(v) for v in m.split('/')) for m in a.modes.split(',')]
    d = torch.device('cuda', 0)
    B = a.batch
    g = torch.Generator(device=d)
    g.manual_seed(0)
    h = lambda *s: (torch.randn(*s, device=d, generator=g) * 0.5).half()
    print('device', torch.cuda.get_device_name(0), 'batch', B, 'modes (impl, job steps)', MODES, flush=True)
    tot = {m: 0.0 for m in MODES}
    for (name, H, W, C, O, K, s, p, dl, cnt) in LAYERS:
        if a.only and a.only not in name:
            continue
        if H == 0:
            N, H, W = B * 300, 1, 1
        else:
            N = B
        Ho, Wo = (H + 2 * p - dl * (K - 1) - 1) // s + 1, (W + 2 * p - dl * (K - 1) - 1) // s + 1
        fl = 2.0 * N * Ho * Wo * O * C * K * K
        Op = (O + 7) // 8 * 8
        per_set = 2 * (N * H * W * C + N * Ho * Wo * Op)
        nbuf = max(1, min(64, -(-a.cold * (1 << 20) // per_set))) if a.cold else 1
        xs, dys = [h(N, H, W, C) for _ in range(nbuf)], [h(N, Ho, Wo, Op) for _ in range(nbuf)]
        if Op != O:
            for t_ in dys:
                t_[..., O:] = 0
        x, dy = xs[0], dys[0]
        ctr = [0]
        cos = torch.arange(0, O, max(1, O // 7), device=d)[:8]
        cis = torch.arange(0, C, max(1, C // 5), device=d)[:6]
        ref = reference_samples(x, dy, K, s, p, dl, Ho, Wo, cos, cis)
        row, base = {}, None
        for m in MODES:
            hip.call('sn_conv_wgrad_impl', m[0], m[1])
            need = hip.query('sn_conv_wgrad_workspace_bytes', N, H, W, C, C, O, Op, K, K, s, p, dl)
            wsb = torch.empty(max(need, 16), dtype=torch.uint8, device=d)
            dw = torch.zeros((O, K * K, C), dtype=torch.float32, device=d)
            def run():
                i = ctr[0] % nbuf
                ctr[0] += 1
                hip.call('sn_conv_wgrad', dys[i], xs[i], dw, N, H, W, C, C, O, Op, K, K, s, p, dl, wsb, need, hip.stream())
            ctr[0] = 0
            run()
            torch.cuda.synchronize()
            got = dw[cos][:, :, cis]
            err = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-6))
            if base is None:
                base, dself = dw.clone(), 0.0
            else:
                dself = float((dw - base).abs().max() / base.abs().max().clamp_min(1e-6))
            us = timeit(run, max(a.iters, 2 * nbuf))
            row[m] = (us, err, dself)
        hip.call('sn_conv_wgrad_impl', 1, 0)
        cells = ' '.join('%s:%7.1f%s' % ('%d/%d' % m, row[m][0], '' if (row[m][1] < 2e-3 and row[m][2] < 2e-3) else '!ERR(ref %.1e self %.1e)' % row[m][1:])
                         for m in MODES)
        bm = min(row, key=lambda m: row[m][0])
        print('%-28s P=%6d Cout=%4d Cin=%5d taps=%d x%2d | %s | best %s %.0f TF/s (legacy %.0f TF/s)' % (
            name, N * Ho * Wo, O, C, K * K, cnt, cells, '%d/%d' % bm, fl / row[bm][0] / 1e6, fl / row[MODES[0]][0] / 1e6), flush=True)
        for m in MODES:
            tot[m] += row[m][0] * cnt
    print('per-step totals (ms): ' + ' '.join('%d/%d:%.2f' % (m[0], m[1], tot[m] / 1e3) for m in MODES))


if __name__ == '__main__':
    main()
