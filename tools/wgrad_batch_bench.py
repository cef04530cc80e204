"""Batched weight gradient (sn_conv_wgrad_batch, csrc/conv_wgrad_ps.hip) against per-layer launches, R101 / batch-20 shapes:
every group is checked against the per-layer result of the gather fallback kernel (impl 0) and timed with distinct operands per layer.

    python tools/wgrad_batch_bench.py [--iters 10]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sniper_amd import hip  # noqa: E402
from tools.wgrad_tune import LAYERS, timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=20)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--job-steps', type=int, default=0, help='+100000 / +200000: force 128- / 256-row tiles')
    ap.add_argument('--trace', action='store_true', help='phase cycles per job (median over jobs) of the batched launch')
    ap.add_argument('--only', default='')
    a = ap.parse_args()
    d = torch.device('cuda', 0)
    g = torch.Generator(device=d)
    g.manual_seed(0)
    h = lambda *s: (torch.randn(*s, device=d, generator=g) * 0.5).half()
    L = {l[0]: l for l in LAYERS}

    def make(name, copies):
        (_, H, W, C, O, K, s, p, dl, _) = L[name]
        N = a.batch
        if H == 0:
            N, H, W = a.batch * 300, 1, 1
        Ho, Wo = (H + 2 * p - dl * (K - 1) - 1) // s + 1, (W + 2 * p - dl * (K - 1) - 1) // s + 1
        Op = (O + 7) // 8 * 8
        out = []
        for _ in range(copies):
            x, dy = h(N, H, W, C), h(N, Ho, Wo, Op)
            if Op != O:
                dy[..., O:] = 0
            dw = torch.zeros((O, K * K, C), dtype=torch.float32, device=d)
            out.append((dy, x, dw, N, H, W, C, C, O, Op, K, K, s, p, dl))
        return out, 2.0 * N * Ho * Wo * O * C * K * K * copies

    groups = [
        ('16 x s3 1x1 1024->256', [('s3 1x1 1024->256 @32', 16)]),
        ('16 x s3 1x1 256->1024', [('s3 1x1 256->1024 @32', 16)]),
        ('7 x s3 3x3 256->256', [('s3 3x3 256->256 @32', 7)]),
        ('6 stage-3 units (18 layers)', [('s3 1x1 1024->256 @32', 6), ('s3 3x3 256->256 @32', 6), ('s3 1x1 256->1024 @32', 6)]),
        ('stage 2 (3 units)', [('s2 1x1 512->128 @64', 3), ('s2 3x3 128->128 @64', 3), ('s2 1x1 128->512 @64', 3)]),
        ('stage 4 tail', [('s4 deform gemm 4608->512', 3), ('s4 1x1 512->2048 @32', 3), ('s4 1x1 2048->512 @32', 2), ('s4 off 3x3d2 512->72', 3)]),
        ('rpn 3x3 alone', [('rpn 3x3 3072->512 @32', 1)]),
        ('heads', [('fc_new_1 12544->1024 x6000', 1), ('fc_new_2 1024->1024 x6000', 1), ('fc cls 1024->81 x6000', 1), ('conv_new_1 1x1 2048->256', 1)]),
    ]
    print('device', torch.cuda.get_device_name(0), 'batch', a.batch, flush=True)
    for gname, spec in groups:
        if a.only and a.only not in gname:
            continue
        probs, fl = [], 0.0
        for name, copies in spec:
            pr, f = make(name, copies)
            probs += pr
            fl += f
        # reference: per-layer launches of the gather kernel (impl 0)
        hip.call('sn_conv_wgrad_impl', 0, 0)
        need0 = max(hip.query('sn_conv_wgrad_workspace_bytes', *pr[3:]) for pr in probs)
        ws0 = torch.empty(max(need0, 16), dtype=torch.uint8, device=d)

        def per_layer():
            for pr in probs:
                hip.call('sn_conv_wgrad', pr[0], pr[1], pr[2], *pr[3:], ws0, need0, hip.stream())
        for pr in probs:
            pr[2].zero_()
        per_layer()
        torch.cuda.synchronize()
        ref = [pr[2].clone() for pr in probs]
        t_old = timeit(per_layer, a.iters)
        # per-layer launches of the new kernel
        hip.call('sn_conv_wgrad_impl', 1, a.job_steps)
        need1 = max(hip.query('sn_conv_wgrad_workspace_bytes', *pr[3:]) for pr in probs)
        ws1 = torch.empty(max(need1, 16), dtype=torch.uint8, device=d)

        def per_layer_new():
            for pr in probs:
                hip.call('sn_conv_wgrad', pr[0], pr[1], pr[2], *pr[3:], ws1, need1, hip.stream())
        for pr in probs:
            pr[2].zero_()
        per_layer_new()
        torch.cuda.synchronize()
        e1 = max(float((pr[2] - r).abs().max() / r.abs().max().clamp_min(1e-6)) for pr, r in zip(probs, ref))
        t_new1 = timeit(per_layer_new, a.iters)
        # one batched launch
        tab = hip.wgrad_table(probs)
        need = hip.query('sn_conv_wgrad_batch_workspace_bytes', tab, len(probs))
        ws = torch.empty(max(need, 16), dtype=torch.uint8, device=d)

        def batched():
            hip.call('sn_conv_wgrad_batch', tab, len(probs), ws, need, hip.stream())
        for pr in probs:
            pr[2].zero_()
        batched()
        torch.cuda.synchronize()
        e2 = max(float((pr[2] - r).abs().max() / r.abs().max().clamp_min(1e-6)) for pr, r in zip(probs, ref))
        t_b = timeit(batched, a.iters)
        if a.trace:
            tr = torch.zeros((8192, 8), dtype=torch.int64, device=d)
            hip.call('sn_conv_wgrad_trace', tr)
            batched()
            torch.cuda.synchronize()
            hip.call('sn_conv_wgrad_trace', None)
            t = tr[tr[:, 7] > 0].double()
            med = t.median(dim=0).values.tolist()
            names = ['life', 'kloop', 'cons_barrier', 'epilogue', 'prod_vmwait', 'prod_barrier', 'prod_issue', 'ksteps']
            print('   trace (%d jobs, median cycles): ' % t.shape[0] + '  '.join('%s %.0f' % (n, v) for n, v in zip(names, med)) +
                  '  | per K-step: loop %.0f  cons wait %.0f  prod vmwait %.0f  prod barrier %.0f  prod issue %.0f' % tuple(
                      med[k] / max(med[7], 1) for k in (1, 2, 4, 5, 6)), flush=True)
        flag = '' if max(e1, e2) < 2e-3 else '  !ERR'
        print('%-30s %2d layers %7.1f GF | per-layer old %8.1f us (%4.0f TF/s) | per-layer new %8.1f us (%4.0f TF/s) err %.1e | batched %8.1f us '
              '(%4.0f TF/s) err %.1e ws %.0f MB%s' % (gname, len(probs), fl / 1e9, t_old, fl / t_old / 1e6, t_new1, fl / t_new1 / 1e6, e1, t_b,
                                                    fl / t_b / 1e6, e2, need / 1e6, flag), flush=True)
    hip.call('sn_conv_wgrad_impl', 1, 0)


if __name__ == '__main__':
    main()
