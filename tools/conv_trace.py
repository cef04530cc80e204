"""Per-workgroup phase timeline of the LDS-DMA convolution kernel (conv_dma.hip `stamp()`): where a workgroup's lifetime goes --
index prologue + first operand stage in flight, K loop, epilogue stores, statistics -- and how the workgroups of one launch
overlap in time.  The library must be loaded with SNIPER_CONV_TRACE=1 (this script sets it before importing sniper_amd).

    python tools/conv_trace.py [--batch 20] [--cfgs 14,18] [--only 's3 ']
"""
import argparse
import os
import sys

os.environ['SNIPER_CONV_TRACE'] = '1'
import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sniper_amd import hip  # noqa: E402
from conv_tune import LAYERS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=20)
    ap.add_argument('--cfgs', default='14,18')
    ap.add_argument('--only', default='s3 ')
    a = ap.parse_args()
    cfgs = [int(c) for c in a.cfgs.split(',')]
    d = torch.device('cuda', 0)
    B = a.batch
    g = torch.Generator(device=d)
    g.manual_seed(0)
    h = lambda *s: (torch.randn(*s, device=d, generator=g) * 0.5).half()
    trace = torch.zeros(8 * 16384, dtype=torch.int64, device=d)
    cold = [h(64, 1024, 1024) for _ in range(3)]      # 384 MB written between launches: the operands come from HBM
    print('device', torch.cuda.get_device_name(0), 'batch', B)
    for (name, H, W, C, O, K, s, p, dl, nf, nd) in LAYERS:
        if a.only and a.only not in name:
            continue
        if H == 0:
            N, H, W = B * 300, 1, 1
        else:
            N = B
        Ho, Wo = (H + 2 * p - dl * (K - 1) - 1) // s + 1, (W + 2 * p - dl * (K - 1) - 1) // s + 1
        M = N * Ho * Wo
        Op = (O + 7) // 8 * 8
        x, w = h(N, H, W, C), h(O, K * K, C)
        y = torch.empty((N, Ho, Wo, O), dtype=torch.float16, device=d)
        dy, wt, dx = h(N, Ho, Wo, Op), h(C, K * K, Op), torch.empty((N, H, W, C), dtype=torch.float16, device=d)
        part = torch.empty((4096, 2, O), dtype=torch.float32, device=d)
        expand = K == 1 and s == 1 and O == 4 * C
        reduce_ = K == 1 and s == 1 and C == 4 * O
        res = h(N, Ho, Wo, O) if expand else None
        acc = h(N, H, W, C) if reduce_ else None
        for direction, cnt in (('fwd', nf), ('dgrad', nd)):
            if cnt == 0:
                continue

            def run():
                if direction == 'fwd':
                    if hip.query('sn_conv_fwd_stats_blocks', N, H, W, C, C, O, O, O if res is not None else 0, K, K, s, p, dl) > 0:
                        hip.call('sn_conv_fwd_stats', x, w, None, res, y, N, H, W, C, C, O, O, O if res is not None else 0, K, K, s, p, dl, 0,
                                 part, hip.stream())
                    else:
                        hip.call('sn_conv_fwd', x, w, None, None, y, N, H, W, C, C, O, O, O, K, K, s, p, dl, 0, 0, hip.stream())
                else:
                    hip.call('sn_conv_dgrad', dy, wt, acc, dx, N, H, W, C, C, Op, Op, C, K, K, s, p, dl, 0, hip.stream())
            for c in cfgs:
                hip.call('sn_conv_tune', c)
                hip.call('sn_conv_trace', None)
                for _ in range(2):
                    run()
                for warm in (True, False):
                    if not warm:
                        for t in cold:
                            t.add_(1.0)
                    trace.zero_()
                    torch.cuda.synchronize()
                    hip.call('sn_conv_trace', trace)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    run()
                    e1.record()
                    torch.cuda.synchronize()
                    hip.call('sn_conv_trace', None)
                    t = trace.cpu().numpy().reshape(-1, 8)
                    t = t[t[:, 0] > 0]
                    real = (t[:, 6] - t[:, 5]).astype(np.float64)          # 100 MHz ticks of the workgroup's life
                    t = t[:, :5].astype(np.float64)
                    if not len(t):
                        print('%-26s %-5s cfg %d: no stamps (register-staged kernel chosen)' % (name, direction, c))
                        continue
                    # (the shader clocks of different XCDs are not synchronised: only differences inside one workgroup mean anything)
                    ph = np.diff(t, axis=1)          # entry->first stage, K loop, stores drained, stats/exit
                    life = t[:, 4] - t[:, 0]
                    print('%-26s %-5s cfg %2d %s: %4d WGs, event %5.1f us | per WG (median cyc): fill %5.0f  kloop %5.0f  stores %5.0f  '
                          'tail %5.0f  life %6.0f (p90 %6.0f)  clock %4.0f MHz' % (
                              name, direction, c, 'warm' if warm else 'cold', len(t), e0.elapsed_time(e1) * 1e3,
                              np.median(ph[:, 0]), np.median(ph[:, 1]), np.median(ph[:, 2]), np.median(ph[:, 3]), np.median(life),
                              np.percentile(life, 90), np.median(life[real > 0] / real[real > 0]) * 100.0), flush=True)
            hip.call('sn_conv_tune', -1)


if __name__ == '__main__':
    main()
