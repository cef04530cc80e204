"""Per-kernel micro-benchmarks at BASELINE C2 shapes (R101, 20 chips of 512x512): HIP-event timing of
the C-ABI entry points on torch's current stream.  Prints one line per kernel with achieved TFLOP/s
(MFMA-bound ops) or GB/s (HBM-bound ops).  Used under rocprofv3 to produce profiles/."""
import argparse
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sniper_amd import hip  # noqa: E402


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters  # ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--quick', action='store_true')
    ap.add_argument('--batch', type=int, default=20)
    ap.add_argument('--only', default='', help='run only the conv shapes whose name contains this, then exit')
    a = ap.parse_args()
    d = torch.device('cuda', 0)
    B = a.batch
    it = 3 if a.quick else 10
    h = lambda *s: (torch.randn(*s, device=d) * 0.5).half()
    print('device', torch.cuda.get_device_name(0), 'batch', B, flush=True)
    convs = [
        # name, H, W, Cin, Cout, K, stride, pad, dil
        ('stage1 1x1 64->256  @128', 128, 128, 64, 256, 1, 1, 0, 1),
        ('stage1 3x3 64->64   @128', 128, 128, 64, 64, 3, 1, 1, 1),
        ('stage2 3x3 128->128 @64', 64, 64, 128, 128, 3, 1, 1, 1),
        ('stage2 1x1 128->512 @64', 64, 64, 128, 512, 1, 1, 0, 1),
        ('stage3 1x1 1024->256 @32', 32, 32, 1024, 256, 1, 1, 0, 1),
        ('stage3 3x3 256->256 @32', 32, 32, 256, 256, 3, 1, 1, 1),
        ('stage3 1x1 256->1024 @32', 32, 32, 256, 1024, 1, 1, 0, 1),
        ('stage4 1x1 2048->512 @32', 32, 32, 2048, 512, 1, 1, 0, 1),
        ('rpn 3x3 3072->512 @32', 32, 32, 3072, 512, 3, 1, 1, 1),
    ]
    if a.only:
        convs = [c for c in convs if a.only in c[0]] + [
            ('probe 1x1 4096->512 M=16384', 32, 32, 4096, 512, 1, 1, 0, 1)]   # B=16: exactly 512 tiles
    for name, H, W, C, O, K, s, p, dl in convs:
        if name.startswith('probe'):
            B = 16
        x, w = h(B, H, W, C), h(O, K * K, C)
        Ho, Wo = (H + 2 * p - dl * (K - 1) - 1) // s + 1, (W + 2 * p - dl * (K - 1) - 1) // s + 1
        y = torch.empty((B, Ho, Wo, O), dtype=torch.float16, device=d)
        fl = 2.0 * B * Ho * Wo * O * C * K * K
        ms = timeit(lambda: hip.call('sn_conv_fwd', x, w, None, None, y, B, H, W, C, C, O, O, O, K, K, s, p, dl, 0, 0, hip.stream()), it)
        print('conv_fwd  %-28s %8.3f ms %8.1f TFLOP/s' % (name, ms, fl / ms / 1e9), flush=True)
        if H <= 64:
            wt, dx = h(C, K * K, O), torch.empty_like(x)
            ms = timeit(lambda: hip.call('sn_conv_dgrad', y, wt, None, dx, B, H, W, C, C, O, O, C, K, K, s, p, dl, 0, hip.stream()), it)
            print('conv_dgrad %-27s %8.3f ms %8.1f TFLOP/s' % (name, ms, fl / ms / 1e9), flush=True)
            dw = torch.zeros((O, K * K, C), dtype=torch.float32, device=d)
            need = hip.query('sn_conv_wgrad_workspace_bytes', B, H, W, C, C, O, O, K, K, s, p, dl)
            wsb = torch.empty(max(need, 16), dtype=torch.uint8, device=d)
            ms = timeit(lambda: hip.call('sn_conv_wgrad', y, x, dw, B, H, W, C, C, O, O, K, K, s, p, dl, wsb, need, hip.stream()), it)
            print('conv_wgrad %-27s %8.3f ms %8.1f TFLOP/s' % (name, ms, fl / ms / 1e9), flush=True)
    if a.only:
        return
    # fc_new_1: 6000 x 12544 -> 1024
    M, K_, O = B * 300, 12544, 1024
    x, w, y = h(M, K_), h(O, K_), torch.empty((M, O), dtype=torch.float16, device=d)
    ms = timeit(lambda: hip.call('sn_conv_fwd', x, w, None, None, y, M, 1, 1, K_, K_, O, O, O, 1, 1, 1, 0, 1, 1, 0, hip.stream()), it)
    print('fc_new_1 fwd %-25s %8.3f ms %8.1f TFLOP/s' % ('%dx%d->%d' % (M, K_, O), ms, 2.0 * M * K_ * O / ms / 1e9), flush=True)
    # BN stage2-size tensor
    for (H, C) in ((64, 512), (32, 1024)):
        Mr = B * H * H
        x, y = h(Mr, C), torch.empty((Mr, C), dtype=torch.float16, device=d)
        ws = torch.empty(hip.query('sn_bn_workspace_bytes', Mr, C), dtype=torch.uint8, device=d)
        f = lambda: torch.ones(C, device=d)
        sc, sh, mu, iv, g, bt = f(), f(), f(), f(), f(), f()
        ms = timeit(lambda: hip.call('sn_bn_stats', x, Mr, C, C, ws, hip.stream()), it)
        print('bn_stats  %dx%d %8.3f ms %8.1f GB/s' % (Mr, C, ms, Mr * C * 2 / ms / 1e6), flush=True)
        ms = timeit(lambda: hip.call('sn_bn_apply', x, y, Mr, C, C, C, sc, sh, 1, hip.stream()), it)
        print('bn_apply  %dx%d %8.3f ms %8.1f GB/s' % (Mr, C, ms, Mr * C * 4 / ms / 1e6), flush=True)
        dg, db = f(), f()
        ms = timeit(lambda: hip.call('sn_bn_backward', y, x, None, y, Mr, C, C, C, C, C, sc, sh, mu, iv, 1, ws, dg, db, hip.stream()), it)
        print('bn_bwd    %dx%d %8.3f ms %8.1f GB/s (10 B/elt)' % (Mr, C, ms, Mr * C * 10 / ms / 1e6), flush=True)
    # NMS 6000 x B
    rs = np.random.RandomState(0)
    c = rs.uniform(0, 512, (B, 6000, 2))
    wh = np.exp(rs.uniform(np.log(8), np.log(300), (B, 6000, 2)))
    dets = np.concatenate((c - wh / 2, c + wh / 2, np.sort(rs.uniform(0, 1, (B, 6000, 1)), 1)[:, ::-1]), 2).astype(np.float32)
    dd = torch.from_numpy(dets).to(d)
    from sniper_amd.ext import gpu_nms
    ms = timeit(lambda: gpu_nms.nms_sorted_device(dd, 0.7, 300), it)
    print('nms B=%d N=6000 keep 300      %8.3f ms' % (B, ms), flush=True)
    ms = timeit(lambda: gpu_nms.nms_sorted_device(dd, 0.7, 0), it)
    print('nms B=%d N=6000 full          %8.3f ms  mask %.1f MB -> %.1f GB/s' % (B, ms, B * 6000 * 94 * 8 / 2 / 1e6, B * 6000 * 94 * 8 / 2 / ms / 1e6), flush=True)
    # proposal target
    A, F = 21, 32
    logits = torch.randn(B, 2, A * F, F, device=d)
    cls_prob = torch.softmax(logits, 1).contiguous()
    bbox_pred = (torch.randn(B, 4 * A, F, F, device=d) * 0.2).contiguous()
    im_info = torch.tensor([[512, 512, 1.6]] * B, device=d)
    gt = -torch.ones(B, 100, 5, device=d)
    gt[:, :5] = torch.tensor([[50, 60, 200, 220, 3], [300, 100, 420, 260, 7], [10, 300, 120, 480, 9], [200, 200, 260, 280, 1], [400, 400, 500, 500, 5]], device=d)
    vr = torch.tensor([[0, 512.0]] * B, device=d)
    from sniper_amd.data.anchors import generate_anchors
    base = torch.from_numpy(generate_anchors(16, (0.5, 1, 2), np.array((2, 4, 7, 10, 13, 16, 24), np.float32)).astype(np.float32)).to(d)
    ws = torch.empty(hip.query('sn_proposal_workspace_bytes', B, A, F, F, 6000, 300), dtype=torch.uint8, device=d)
    rois, lab = torch.empty(B * 300, 5, device=d), torch.empty(B * 300, device=d)
    tg, wg = torch.empty(B * 300, 4, device=d), torch.empty(B * 300, 4, device=d)
    stds = np.array([0.1, 0.1, 0.2, 0.2], np.float32)
    ms = timeit(lambda: hip.call('sn_multi_proposal_target', cls_prob, bbox_pred, im_info, gt, vr, base, B, A, F, F, 16, 100, 6000, 300, 0.7, 0.0,
                                 0.5, stds.ctypes.data, ws, rois, lab, tg, wg, hip.stream()), it)
    print('multi_proposal_target B=%d     %8.3f ms' % (B, ms), flush=True)
    # anchor assignment on golden-like chips
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
    from golden_util import anchor_case, ref_cfg
    from sniper_amd.data.anchors import AnchorAssigner
    aa = AnchorAssigner(ref_cfg(), 512)
    chips = [anchor_case(k % 42)[0] for k in range(B)]
    ms = timeit(lambda: aa.assign(chips, seed=1), it)
    print('anchor_assign B=%d (incl. host packing) %8.3f ms -> %.0f chips/s' % (B, ms, B / ms * 1e3), flush=True)
    # DPSROI
    R, C = B * 300, 256
    data = h(B, 32, 32, C)
    out = torch.empty((R, 7, 7, C), dtype=torch.float16, device=d)
    ms = timeit(lambda: hip.call('sn_dpsroi_pool_fwd', data, rois, None, out, R, 32, 32, C, 7, 4, 1 / 16., 0.0, hip.stream()), it)
    print('dpsroi fwd R=%d               %8.3f ms %8.1f GB/s (output bytes)' % (R, ms, R * 49 * C * 2 / ms / 1e6), flush=True)
    dda = torch.empty((B, 32, 32, C), dtype=torch.float16, device=d)
    wsb = torch.empty(hip.query('sn_dpsroi_bwd_workspace_bytes', R), dtype=torch.uint8, device=d)
    ms = timeit(lambda: hip.call('sn_dpsroi_pool_bwd', out, data, rois, None, dda, 0, None, R, B, 32, 32, C, 7, 4, 1 / 16., 0.0, wsb,
                                 hip.stream()), it)
    print('dpsroi bwd R=%d (no trans)    %8.3f ms %8.1f GB/s (dout bytes)' % (R, ms, R * 49 * C * 2 / ms / 1e6), flush=True)
    trans = torch.randn(R, 2, 7, 7, device=d) * 0.3
    dtr = torch.empty_like(trans)
    ms = timeit(lambda: hip.call('sn_dpsroi_pool_bwd', out, data, rois, trans, dda, 0, dtr, R, B, 32, 32, C, 7, 4, 1 / 16., 0.1, wsb,
                                 hip.stream()), it)
    print('dpsroi bwd R=%d (trans)       %8.3f ms' % (R, ms), flush=True)
    # deformable conv sampling (res5: 512 ch, 3x3 dil 2, 4 groups)
    C5, DG = 512, 4
    x5 = h(B, 32, 32, C5)
    off = (torch.randn(B, 32, 32, 72, device=d) * 0.5).half()
    col = torch.empty((B * 1024, 9 * C5), dtype=torch.float16, device=d)
    ms = timeit(lambda: hip.call('sn_deform_im2col', x5, off, col, B, 32, 32, C5, 3, 3, 1, 2, 2, DG, 72, 0, hip.stream()), it)
    print('deform im2col B=%d            %8.3f ms %8.1f GB/s (col bytes)' % (B, ms, col.numel() * 2 / ms / 1e6), flush=True)
    dx5 = torch.empty_like(x5)
    doff = torch.empty_like(off)
    dws5 = torch.zeros(16, dtype=torch.uint8, device=d)
    ms = timeit(lambda: hip.call('sn_deform_col2im', col, x5, off, dx5, 0, doff, B, 32, 32, C5, 3, 3, 1, 2, 2, DG, 72, 0, dws5, hip.stream()), it)
    print('deform col2im B=%d            %8.3f ms %8.1f GB/s (col bytes)' % (B, ms, col.numel() * 2 / ms / 1e6), flush=True)

if __name__ == '__main__':
    main()
