"""-m gpu: the hot kernels at the LAUNCH SHAPES of BASELINE configs[1] (ResNet-101, 20 chips of 512 x 512 per GPU) against
the fp32 CPU oracle, element by element -- not properties (tests/test_gpu_fullsize_properties.py), not small tiles
(tests/test_gpu_nn_ops.py).  These are the shapes whose kernel selection differs from the small cases: the measured
per-layer LDS-DMA tile configurations (conv_dma_choice), the weight gradient's K-splits over 20 480 / 81 920 pixels with
slab reduction, the all-taps 3x3 weight gradient of the RPN, 6000-RoI deformable PS-RoI pooling.
Tolerance everywhere: 1e-2 relative to the tensor's scale (north_star: conv / loss tensors within 1e-2 rel, fp16 storage)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import assert_close, dev, f16r  # noqa: E402
from oracle import nn as onn  # noqa: E402
from test_gpu_nn_ops import _check_dgrad_wgrad, _conv_fwd, _ref_conv  # noqa: E402

B = 20
# N, C, H, W, O, K, stride, pad, dil, bias, res, relu  (the tuple layout of tests/test_gpu_nn_ops.py::CONV_CASES)
C2_CONV_SHAPES = {
    'stage3 3x3 256->256 @32': (B, 256, 32, 32, 256, 3, 1, 1, 1, False, False, 0),
    'stage3 1x1 256->1024 @32 (+residual)': (B, 256, 32, 32, 1024, 1, 1, 0, 1, False, True, 0),
    'stage3 1x1 1024->256 @32': (B, 1024, 32, 32, 256, 1, 1, 0, 1, False, False, 0),
    'stage3 unit1 3x3/2 256->256 @64': (B, 256, 64, 64, 256, 3, 2, 1, 1, False, False, 0),
    'stage2 3x3 128->128 @64': (B, 128, 64, 64, 128, 3, 1, 1, 1, False, False, 0),
    'stage4 offset 3x3 d2 512->72 @32': (B, 512, 32, 32, 72, 3, 1, 2, 2, True, False, 0),
    'rpn 3x3 3072->512 @32 (+bias, relu)': (B, 3072, 32, 32, 512, 3, 1, 1, 1, True, False, 1),
}


@pytest.mark.parametrize('name', list(C2_CONV_SHAPES))
def test_conv_fwd_dgrad_wgrad_at_c2_launch_shapes(name):
    case = C2_CONV_SHAPES[name]
    N, C, H, W, O, K, s, p, d, hb, hr, relu = case
    rs = np.random.RandomState(len(name) * 7 + C)
    x = rs.standard_normal((N, C, H, W)).astype(np.float32)
    w = (rs.standard_normal((O, C, K, K)) / np.sqrt(C * K * K)).astype(np.float32)
    b = rs.standard_normal(O).astype(np.float32) if hb else None
    want0 = _ref_conv(x, w, b, None, s, p, d, 0)
    res = rs.standard_normal(want0.shape).astype(np.float32) if hr else None
    want = _ref_conv(x, w, b, res, s, p, d, relu)
    got = _conv_fwd(x, w, b, res, s, p, d, relu, 0)
    assert_close(got, want, 1e-2, 1e-2 * np.abs(want).max(), 'conv fwd %s' % name)
    _check_dgrad_wgrad(case)


def test_fc_new_1_at_c2_shape():
    """fc_new_1 (resnet_mx_101_e2e.py:288-303): 6000 RoIs x 12544 -> 1024, forward + data gradient + weight gradient."""
    _check_dgrad_wgrad((B * 300, 12544, 1, 1, 1024, 1, 1, 0, 1, False, False, 0))
    case = (B * 300, 12544, 1, 1, 1024, 1, 1, 0, 1, True, False, 1)
    rs = np.random.RandomState(5)
    x = rs.standard_normal((B * 300, 12544, 1, 1)).astype(np.float32)
    w = (rs.standard_normal((1024, 12544, 1, 1)) / np.sqrt(12544)).astype(np.float32)
    b = rs.standard_normal(1024).astype(np.float32)
    want = _ref_conv(x, w, b, None, 1, 0, 1, 1)
    assert_close(_conv_fwd(x, w, b, None, 1, 0, 1, 1, 0), want, 1e-2, 1e-2 * np.abs(want).max(), 'fc_new_1 fwd')


@pytest.mark.parametrize('with_trans', [False, True])
def test_deformable_psroi_pooling_at_6000_rois(with_trans):
    """DeformablePSROIPooling (resnet_mx_101_e2e.py:286-293) at the C2 size: 20 chips x 300 RoIs on the (20, 256, 32, 32) map,
    forward, data gradient and offset gradient against oracle/nn.py (sparse-operator form, pinned to the loop definition by
    tests/test_oracle_graph_cpu.py)."""
    from sniper_amd import hip
    rs = np.random.RandomState(31)
    C, H, W, P, S, SC, R = 256, 32, 32, 7, 4, 16, B * 300
    data = rs.standard_normal((B, C, H, W)).astype(np.float32)
    rois = np.zeros((R, 5), np.float32)
    rois[:, 0] = np.repeat(np.arange(B), 300)
    c = rs.uniform(0, 512, (R, 2))
    wh = np.exp(rs.uniform(np.log(8), np.log(400), (R, 2)))
    rois[:, 1:3], rois[:, 3:5] = np.clip(c - wh / 2, 0, 511), np.clip(c + wh / 2, 0, 511)
    trans = (rs.standard_normal((R, 2, P, P)) * 0.5).astype(np.float32) if with_trans else None
    tstd = 0.1 if with_trans else 0.0
    dd = torch.from_numpy(np.ascontiguousarray(data.transpose(0, 2, 3, 1))).to(dev()).half()
    td = lambda z: torch.from_numpy(z).to(dev())
    out = torch.empty((R, P, P, C), dtype=torch.float16, device=dev())
    hip.call('sn_dpsroi_pool_fwd', dd, td(rois), None if trans is None else td(trans), out, R, H, W, C, P, S, 1.0 / SC, tstd, hip.stream())
    want = onn.dpsroi_pool_fast(f16r(data), rois, trans, P, S, 1.0 / SC, tstd)
    assert_close(out.float().cpu().numpy().transpose(0, 3, 1, 2), want, 1e-2, 1e-2, 'dpsroi fwd R=6000')
    dout = rs.standard_normal((R, C, P, P)).astype(np.float32)
    dod = torch.from_numpy(np.ascontiguousarray(dout.transpose(0, 2, 3, 1))).to(dev()).half()
    wd, wtr = onn.dpsroi_pool_backward_fast(f16r(dout), f16r(data), rois, trans, P, S, 1.0 / SC, tstd)
    ws = torch.empty(hip.query('sn_dpsroi_bwd_workspace_bytes', R), dtype=torch.uint8, device=dev())
    d_data = torch.empty((B, H, W, C), dtype=torch.float16, device=dev())
    d_trans = torch.empty((R, 2, P, P), dtype=torch.float32, device=dev()) if with_trans else None
    hip.call('sn_dpsroi_pool_bwd', dod, dd, td(rois), None if trans is None else td(trans), d_data, 0, d_trans, R, B, H, W, C, P, S,
             1.0 / SC, tstd, ws, hip.stream())
    assert_close(d_data.float().cpu().numpy().transpose(0, 3, 1, 2), wd, 1e-2, 1e-2 * np.abs(wd).max(), 'dpsroi d_data R=6000')
    if with_trans:
        assert_close(d_trans.cpu().numpy(), wtr, 1e-2, 1e-2 * np.abs(wtr).max(), 'dpsroi d_trans R=6000')
