"""Diagnostic sweep for the implicit-GEMM convolution: the R101 stage-2/3/4 shapes at several batch sizes with the 64- and
128-row tile forced (SNIPER_CONV_BM), to separate grid-size effects (tile quantisation, per-launch latency) from the
per-tile efficiency of the kernel.  Prints TFLOP/s per (shape, batch, BM)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sniper_amd import hip  # noqa: E402
from tools.microbench import timeit  # noqa: E402


def main():
    d = torch.device('cuda', 0)
    h = lambda *s: (torch.randn(*s, device=d) * 0.5).half()
    shapes = [
        ('stage2 3x3 128->128 @64', 64, 128, 128, 3),
        ('stage2 1x1 128->512 @64', 64, 128, 512, 1),
        ('stage2 1x1 512->128 @64', 64, 512, 128, 1),
        ('stage3 1x1 1024->256 @32', 32, 1024, 256, 1),
        ('stage3 3x3 256->256 @32', 32, 256, 256, 3),
        ('stage3 1x1 256->1024 @32', 32, 256, 1024, 1),
        ('stage4 1x1 2048->512 @32', 32, 2048, 512, 1),
        ('stage4 1x1 512->2048 @32', 32, 512, 2048, 1),
    ]
    print('%-28s %5s %4s %9s %9s %7s' % ('shape', 'batch', 'BM', 'ms', 'TFLOP/s', 'tiles'))
    for name, H, C, O, K in shapes:
        for B in (20, 40, 80):
            x, w = h(B, H, H, C), h(O, K * K, C)
            y = torch.empty((B, H, H, O), dtype=torch.float16, device=d)
            fl = 2.0 * B * H * H * O * C * K * K
            for bm in ('64', '128'):
                os.environ['SNIPER_CONV_BM'] = bm
                ms = timeit(lambda: hip.call('sn_conv_fwd', x, w, None, None, y, B, H, H, C, C, O, O, O, K, K, 1, K // 2, 1, 0, 0,
                                             hip.stream()), 20, 3)
                tiles = -(-B * H * H // int(bm)) * -(-O // 128)
                print('%-28s %5d %4s %9.4f %9.1f %7d' % (name, B, bm, ms, fl / ms / 1e9, tiles), flush=True)
    os.environ.pop('SNIPER_CONV_BM', None)
    # launch-overhead floor (a 1-element kernel) and the latency chain of a single tile wave (tiny batches)
    one = torch.zeros(8, device=d)
    ms = timeit(lambda: hip.call('sn_ew_f32', one, None, one, 1, 4, 0.0, hip.stream()), 200, 10)
    print('null launch back-to-back: %.2f us' % (ms * 1e3))
    for name, H, C, O, K in (('stage3 1x1 1024->256 @32', 32, 1024, 256, 1), ('stage3 3x3 256->256 @32', 32, 256, 256, 3),
                             ('stage3 1x1 256->1024 @32', 32, 256, 1024, 1)):
        for B in (1, 2, 5, 10, 20, 30, 40, 60, 80):
            x, w = h(B, H, H, C), h(O, K * K, C)
            y = torch.empty((B, H, H, O), dtype=torch.float16, device=d)
            fl = 2.0 * B * H * H * O * C * K * K
            ms = timeit(lambda: hip.call('sn_conv_fwd', x, w, None, None, y, B, H, H, C, C, O, O, O, K, K, 1, K // 2, 1, 0, 0,
                                         hip.stream()), 30, 3)
            print('%-28s B=%3d  %8.2f us %8.1f TFLOP/s' % (name, B, ms * 1e3, fl / ms / 1e9), flush=True)


if __name__ == '__main__':
    main()
