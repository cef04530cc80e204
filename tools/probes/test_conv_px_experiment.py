"""GPU equality tests of the pixel-stationary 1 x 1 EXPERIMENT (tools/probes/conv_px_experiment.hip) against the shipped tile kernel: outputs
and BatchNorm partials bit for bit.  Not collected by the suite (the kernel is not in the shipped library):

    tools/probes/conv_px_build.sh && SNIPER_HIP_LIB=sniper_amd/lib/libsniper_hip_px.so python -m pytest tools/probes/test_conv_px_experiment.py -q
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from gpu_util import assert_close, dev  # noqa: E402


def _hip():
    from sniper_amd import hip
    return hip


def PX(v):
    _hip().lib()._dll.sn_conv_px(int(v))


PX_DEFAULT = int(os.environ.get('SNIPER_CONV_PX', '0') or 0)


@pytest.mark.parametrize('N,H,C,O,extras', [(8, 32, 256, 1024, ''), (5, 41, 128, 512, ''), (9, 32, 192, 384, 'bias res relu'), (11, 28, 256, 128, 'res'),
                                            (20, 32, 256, 1024, '')])
def test_conv_px_forward_equals_the_tile_kernel(N, H, C, O, extras):
    """csrc/conv_px.hip (pixels stationary in registers, weights streamed through LDS) against conv_dma_kernel's 160 x 128 tiles on the
    1 x 1 layers it takes over: output AND the per-160-row BatchNorm partials BIT-equal (same MFMA sequence, same K order, same
    association of the partial sums) -- Cin 128 / 192 / 256, 1 .. 4 chunks of 128 channels, one and two column tiles, ragged last
    row tile, bias / residual / ReLU epilogue, the BASELINE C2 shape."""
    hip = _hip()
    rs = np.random.RandomState(N * H + O)
    M = N * H * H
    xd = torch.from_numpy(rs.standard_normal((N, H, H, C)).astype(np.float32)).to(dev()).half()
    wd = torch.from_numpy((rs.standard_normal((O, 1, C)) / np.sqrt(C)).astype(np.float32)).to(dev()).half()
    rd = torch.from_numpy(rs.standard_normal((N, H, H, O)).astype(np.float32)).to(dev()).half() if 'res' in extras else None
    bias = torch.from_numpy(rs.standard_normal(O).astype(np.float32)).to(dev()) if 'bias' in extras else None
    relu = 1 if 'relu' in extras else 0
    geom = (N, H, H, C, C, O, O, O if rd is not None else 0, 1, 1, 1, 0, 1)
    nblk = hip.query('sn_conv_fwd_stats_blocks', *geom)
    assert nblk == -(-M // 160)
    outs = []
    try:
        for on in (0, 1, 2):              # tile kernel, pixel-stationary, pixel-stationary with the staggered pixel halves
            PX(on)
            assert hip.query('sn_conv_fwd_stats_blocks', *geom) == nblk
            y = torch.full((N, H, H, O), 3.0, dtype=torch.float16, device=dev())
            part = torch.full((nblk, 2, O), 7.0, dtype=torch.float32, device=dev())
            hip.call('sn_conv_fwd_stats', xd, wd, bias, rd, y, *geom, relu, part, hip.stream())
            y2 = torch.full((N, H, H, O), 5.0, dtype=torch.float16, device=dev())
            hip.call('sn_conv_fwd', xd, wd, bias, rd, y2, *geom, relu, 0, hip.stream())
            torch.cuda.synchronize()
            outs.append((y, part, y2))
    finally:
        PX(PX_DEFAULT)
    (y0, p0, z0), (y1, p1, z1), (y2, p2, z2) = outs
    assert torch.equal(y0, y1) and torch.equal(z0, z1) and torch.equal(y0, z0)
    assert torch.equal(p0, p1)
    assert torch.equal(y0, y2) and torch.equal(z0, z2) and torch.equal(p0, p2)
    ref = (xd.float().reshape(M, C) @ wd.float().reshape(O, C).t())
    if bias is not None:
        ref = ref + bias
    if rd is not None:
        ref = ref + rd.float().reshape(M, O)
    if relu:
        ref = ref.clamp_min(0)
    assert_close(y1.float().reshape(M, O).cpu().numpy(), ref.cpu().numpy(), 2e-3, 2e-2, 'against the fp32 product')


@pytest.mark.parametrize('N,H,C,O,act,acc', [(8, 32, 1024, 256, 1, False), (5, 41, 512, 128, 2, False), (9, 32, 384, 192, 0, True), (20, 32, 1024, 256, 1, False)])
def test_conv_px_data_gradient_equals_the_tile_kernel(N, H, C, O, act, acc):
    """The same kernel as the data gradient of a 1 x 1 reduction (dx has C channels, dy has O <= 256: the contraction), with the fused
    BatchNorm-backward reduction over bn_x (ReLU / ReLU6 / no mask) and with an accumulated gradient: dx and partials BIT-equal to
    the tile kernel's."""
    hip = _hip()
    rs = np.random.RandomState(N + H + C)
    M = N * H * H
    dy = torch.from_numpy(rs.standard_normal((N, H, H, O)).astype(np.float32)).to(dev()).half()
    wt = torch.from_numpy((rs.standard_normal((C, 1, O)) / np.sqrt(O)).astype(np.float32)).to(dev()).half()
    bnx = torch.from_numpy(rs.standard_normal((N, H, H, C)).astype(np.float32)).to(dev()).half()
    accd = torch.from_numpy(rs.standard_normal((N, H, H, C)).astype(np.float32)).to(dev()).half() if acc else None
    f = lambda a: torch.from_numpy(a.astype(np.float32)).to(dev())
    scale, shift, mean = f(rs.uniform(0.5, 1.5, C)), f(rs.uniform(-0.5, 2.5, C)), f(rs.standard_normal(C) * 0.1)
    geom = (N, H, H, C, C, O, O, C if acc else 0, 1, 1, 1, 0, 1)
    nblk = hip.query('sn_conv_dgrad_bn_blocks', *geom)
    assert nblk == -(-M // 160)
    outs = []
    try:
        for on in (0, 1, 2):
            PX(on)
            dx = torch.full((N, H, H, C), 3.0, dtype=torch.float16, device=dev())
            part = torch.full((nblk, 2, C), 7.0, dtype=torch.float32, device=dev())
            hip.call('sn_conv_dgrad_bn', dy, wt, accd, dx, *geom, bnx, C, scale, shift, mean, act, part, hip.stream())
            dx2 = torch.full((N, H, H, C), 5.0, dtype=torch.float16, device=dev())
            hip.call('sn_conv_dgrad', dy, wt, accd, dx2, *geom, 0, hip.stream())
            torch.cuda.synchronize()
            outs.append((dx, part, dx2))
    finally:
        PX(PX_DEFAULT)
    (a0, p0, b0), (a1, p1, b1), (a2, p2, b2) = outs
    assert torch.equal(a0, a1) and torch.equal(b0, b1) and torch.equal(a0, b0)
    assert torch.equal(p0, p1)
    assert torch.equal(a0, a2) and torch.equal(b0, b2) and torch.equal(p0, p2)
    assert float(p1[:, 0].abs().sum()) > 0 and float(p1[:, 1].abs().sum()) > 0


