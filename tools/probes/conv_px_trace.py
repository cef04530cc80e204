"""Where a launch of the pixel-stationary 1 x 1 kernel (csrc/conv_px.hip) spends its time: (1) the launch with PARTS of the kernel
switched off (sn_conv_px bits 4+: output stores, weight DMA after the first stage, pixel loads, MFMAs, statistics -- wrong
results, timing only), (2) the phase stamps of every workgroup's wave 0 (shader clock: entry, operands landed, chunk multiplied /
stored, exit).  The library must be loaded with SNIPER_CONV_TRACE=1 (set here).

    tools/probes/conv_px_build.sh && SNIPER_HIP_LIB=sniper_amd/lib/libsniper_hip_px.so python tools/probes/conv_px_trace.py

(round 5 experiment: the kernel is NOT part of the shipped library; tools/probes/conv_px_experiment.hip, profiles/r05_conv_px_experiment.txt)
"""
import os
import sys

os.environ['SNIPER_CONV_TRACE'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from sniper_amd import hip  # noqa: E402


def PX(v):
    """sn_conv_px of the EXPERIMENT library (not in include/sniper_hip.h: called on the ctypes handle)"""
    hip.lib()._dll.sn_conv_px(int(v))

dev = torch.device('cuda:0')
SETS = 4
ITERS = 40


def rnd(rs, *shape, scale=1.0):
    return torch.from_numpy((rs.standard_normal(shape) * scale).astype(np.float32)).to(dev).half()


def timed(fn, iters=ITERS):
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


def case(kind, N, H, C, O):
    """kind 'fwd': x (C channels) -> y (O channels) + statistics; 'dgrad': dx (C channels) <- dy (O channels) + BatchNorm-backward reduction"""
    rs = np.random.RandomState(C + O)
    M = N * H * H
    if kind == 'fwd':
        xs = [rnd(rs, N, H, H, C) for _ in range(SETS)]
        w = rnd(rs, O, 1, C, scale=1.0 / np.sqrt(C))
        geom = (N, H, H, C, C, O, O, 0, 1, 1, 1, 0, 1)
        nblk = hip.query('sn_conv_fwd_stats_blocks', *geom)
        ys = [torch.empty((N, H, H, O), dtype=torch.float16, device=dev) for _ in range(SETS)]
        ps = [torch.zeros((nblk, 2, O), dtype=torch.float32, device=dev) for _ in range(SETS)]
        run = lambda k: hip.call('sn_conv_fwd_stats', xs[k], w, None, None, ys[k], *geom, 0, ps[k], hip.stream())
        name = 'sn_conv_fwd_stats N%d %dx%d C%d->%d' % (N, H, H, C, O)
    else:
        dys = [rnd(rs, N, H, H, O) for _ in range(SETS)]
        bnxs = [rnd(rs, N, H, H, C) for _ in range(SETS)]
        wt = rnd(rs, C, 1, O, scale=1.0 / np.sqrt(O))
        f = lambda a: torch.from_numpy(a.astype(np.float32)).to(dev)
        scale, shift, mean = f(rs.uniform(0.5, 1.5, C)), f(rs.uniform(-0.5, 2.5, C)), f(rs.standard_normal(C) * 0.1)
        geom = (N, H, H, C, C, O, O, 0, 1, 1, 1, 0, 1)
        nblk = hip.query('sn_conv_dgrad_bn_blocks', *geom)
        dxs = [torch.empty((N, H, H, C), dtype=torch.float16, device=dev) for _ in range(SETS)]
        ps = [torch.zeros((nblk, 2, C), dtype=torch.float32, device=dev) for _ in range(SETS)]
        run = lambda k: hip.call('sn_conv_dgrad_bn', dys[k], wt, None, dxs[k], *geom, bnxs[k], C, scale, shift, mean, 1, ps[k], hip.stream())
        name = 'sn_conv_dgrad_bn N%d %dx%d dx C%d <- dy C%d' % (N, H, H, C, O)
    print('== ' + name, flush=True)
    outs = ys if kind == 'fwd' else dxs
    hip.call('sn_conv_trace', None)
    PX(0)
    run(0)
    torch.cuda.synchronize()
    ref_o, ref_p = outs[0].clone(), ps[0].clone()
    for mode in (1, 2):
        PX(mode)
        outs[0].fill_(3.0)
        ps[0].fill_(7.0)
        run(0)
        torch.cuda.synchronize()
        print('   px %d bit-equal to the tile kernel: output %s, partials %s' % (mode, torch.equal(outs[0], ref_o), torch.equal(ps[0], ref_p)), flush=True)
    hip.call('sn_conv_trace', None)
    PX(0)
    print('   tile kernel                         %6.1f us' % timed(run), flush=True)
    for mode in (1, 2):
        for dbg, what in ((0, 'complete'), (1, 'no output stores'), (2, 'no weight DMA after stage 0'), (4, 'no pixel loads'), (8, 'no MFMAs'),
                          (16, 'no statistics'), (1 | 16, 'no stores, no statistics'), (1 | 2 | 16, 'no stores / DMA / statistics'),
                          (1 | 2 | 4 | 16, 'MFMAs + fragment reads only'), (1 | 2 | 4 | 8 | 16, 'skeleton (barriers, LDS reads)')):
            PX(mode | (dbg << 4))
            print('   px %d %-32s %6.1f us' % (mode, what, timed(run)), flush=True)
    # phase stamps of one launch (warm: the 5th of a row), both variants
    trace = torch.zeros(16 * 4096, dtype=torch.int64, device=dev)
    for mode in (1, 2):
        PX(mode)
        for k in range(SETS):
            run(k)
        trace.zero_()
        torch.cuda.synchronize()
        hip.call('sn_conv_trace', trace)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(0)
        e1.record()
        torch.cuda.synchronize()
        hip.call('sn_conv_trace', None)
        t = trace.cpu().numpy().reshape(-1, 16)
        t = t[t[:, 0] > 0].astype(np.float64)
        if not len(t):
            print('   px %d: no stamps' % mode)
            continue
        life = t[:, 10] - t[:, 0]
        cols = [k for k in range(1, 11) if (t[:, k] > 0).all()]
        rel = {k: np.median(t[:, k] - t[:, 0]) for k in cols}
        print('   px %d stamps: %d workgroups, event %.1f us, life median %.0f p90 %.0f ticks (%.3f us / tick if life == event); since entry (median ticks): %s'
              % (mode, len(t), e0.elapsed_time(e1) * 1e3, np.median(life), np.percentile(life, 90), e0.elapsed_time(e1) * 1e3 / max(np.median(life), 1),
                 '  '.join('[%d] %.0f' % (k, rel[k]) for k in cols)), flush=True)
    PX(int(os.environ.get('SNIPER_CONV_PX', '0') or 0))


if __name__ == '__main__':
    case('fwd', 20, 32, 256, 1024)
    case('dgrad', 20, 32, 1024, 256)
    case('fwd', 20, 64, 128, 512)
