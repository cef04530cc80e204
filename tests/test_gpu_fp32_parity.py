"""-m gpu: the layers the reference runs in FP32 -- everything behind its Cast(float32) (symbols/faster/resnet_mx_101_e2e.py:250-252):
rpn_conv_3x3 -> rpn_cls_score / rpn_bbox_pred (:147-155) and fc_new_1 -> fc_new_2 -> cls_score / bbox_pred (:288-303) -- against an
oracle that is fed the UN-ROUNDED fp32 weights, activations and output gradients (torch CPU, fp32 arithmetic), at the BASELINE
configs[1] shapes.  Every other convolution test rounds the operands to fp16 first (gpu_util.f16r), so it measures the kernels'
arithmetic only; this one measures what the fp16 storage of this engine (MFMA operands, the intermediate activation, the
incoming gradient) costs against the reference's fp32 arithmetic.  north_star: conv / loss tensors within 1e-2 relative, fp16.
Asserted per tensor: every element within 1e-2 of the tensor's largest magnitude, relative L2 error <= 3e-3, and EVERY significant
element (at least 5 % of the largest magnitude) within 1e-2 of ITS OWN magnitude; the quantiles over all elements down to 1e-3
of the scale (where fp16 storage's absolute error of ~1e-4 of the scale shows as a few percent relative) are printed and go to
gpurun_out/parity_quantiles.jsonl (gpu_util.assert_close).
ReLU decisions are teacher-forced in the BACKWARD pass only: the oracle's forward uses its own relu (and is compared as such), its
backward uses the device's mask.  A pre-activation within ~5e-4 of zero (the size of the fp16 storage error there) lands on the
other side of zero for ~3e-4 of the elements; each such flip switches a full-size gradient term on or off, which on this random
data adds noise of ~2 % of the weight gradient's RMS -- a property of ReLU under ANY rounding (the fp32 reference flips against
fp64 the same way, just less often), not of the arithmetic being measured here (same method as tests/test_gpu_engine.py's
teacher-forced end-to-end runs; first measured without it: 0.4 % of rpn_conv_3x3's dW elements beyond 1e-2 of the scale)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as Fnn

pytestmark = pytest.mark.gpu

from gpu_util import assert_close, dev, error_quantiles, to_nhwc_f16, from_nhwc, w_to_otI  # noqa: E402

B = 20


def _hip():
    from sniper_amd import hip
    return hip


def _pad8(n):
    return (n + 7) // 8 * 8


class _Layer(object):
    """one Convolution / FullyConnected on the device through the C ABI, fp16 storage"""

    def __init__(self, w32, b32, k, pad, relu, out_f32):
        hip = _hip()
        self.O, self.C = w32.shape[0], w32.shape[1]
        self.k, self.pad, self.relu, self.out_f32 = k, pad, relu, out_f32
        self.w = torch.from_numpy(w_to_otI(w32)).to(dev()).half().contiguous()
        self.b = torch.from_numpy(b32).to(dev())
        self.Op = _pad8(self.O)
        wm = torch.from_numpy(w_to_otI(w32)).to(dev())
        self.wT = torch.empty((self.C, k * k, self.Op), dtype=torch.float16, device=dev())
        hip.call('sn_weight_transpose', wm, self.wT, self.O, k * k, self.C, self.Op, hip.stream())

    def fwd(self, x):
        hip = _hip()
        N, H, W, C = x.shape
        self.x = x
        y = torch.empty((N, H, W, self.O), dtype=torch.float32 if self.out_f32 else torch.float16, device=dev())
        hip.call('sn_conv_fwd', x, self.w, self.b, None, y, N, H, W, C, C, self.O, self.O, self.O, self.k, self.k, 1, self.pad, 1,
                 self.relu, 1 if self.out_f32 else 0, hip.stream())
        self.y = y
        return y

    def bwd(self, dy, dx_acc=None, want_dx=True):
        """dy: device (N,H,W,Op) fp16 (what the executor hands a convolution: ops._GemmLike.dy_act) -> (dx fp16, dw fp32)"""
        hip = _hip()
        N, H, W, C = self.x.shape
        dw = torch.zeros((self.O, self.k * self.k, C), dtype=torch.float32, device=dev())
        need = hip.query('sn_conv_wgrad_workspace_bytes', N, H, W, C, C, self.O, self.Op, self.k, self.k, 1, self.pad, 1)
        ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev())
        hip.call('sn_conv_wgrad', dy, self.x, dw, N, H, W, C, C, self.O, self.Op, self.k, self.k, 1, self.pad, 1, ws, need, hip.stream())
        dx = None
        if want_dx:
            dx = dx_acc if dx_acc is not None else torch.empty((N, H, W, C), dtype=torch.float16, device=dev())
            hip.call('sn_conv_dgrad', dy, self.wT, dx_acc, dx, N, H, W, C, C, self.Op, self.Op, C, self.k, self.k, 1, self.pad, 1, 0,
                     hip.stream())
        return dx, dw


def _dy16(g32_nchw, Op):
    """fp32 gradient of an fp32 output -> channels-last fp16 with the channel pitch padded to 8 (zeros)"""
    N, O, H, W = g32_nchw.shape
    t = torch.zeros((N, H, W, Op), dtype=torch.float16, device=dev())
    t[..., :O] = to_nhwc_f16(g32_nchw)
    return t


def _relu_bwd(g, y):
    hip = _hip()
    out = torch.empty_like(g)
    C = g.shape[-1]
    hip.call('sn_ew_f16', g, None, y, out, g.numel() // C, C, C, C, C, C, 2, hip.stream())
    return out


def _dw_oihw(dw, k):
    O, _, C = dw.shape
    return dw.cpu().numpy().reshape(O, k, k, C).transpose(0, 3, 1, 2)


def _check(what, got, want):
    assert_close(got, want, 0.0, 1e-2 * float(np.abs(want).max()), what)      # every element within 1e-2 of the tensor's scale
    q = error_quantiles(got, want)
    assert q['rel_l2'] <= 3e-3, (what, q)
    assert q['rel_significant']['p100'] <= 1e-2, (what, q)
    print('%-24s n %9d  rel L2 %.1e | significant elements: rel p50 %.1e p99 %.1e max %.1e | all >= 1e-3 of scale: rel p50 %.1e p99 %.1e '
          'p99.9 %.1e | err/max p99.9 %.1e max %.1e' % (
              what, q['n'], q['rel_l2'], q['rel_significant']['p50'], q['rel_significant']['p99'], q['rel_significant']['p100'],
              q['rel_elementwise']['p50'], q['rel_elementwise']['p99'], q['rel_elementwise']['p99.9'], q['err_over_max']['p99.9'],
              q['err_over_max']['p100']))


def test_rpn_head_against_unrounded_fp32():
    """rpn_conv_3x3 (3072 -> 512, +bias, ReLU) -> rpn_cls_score (1x1, 42) and rpn_bbox_pred (1x1, 84) on the (20, 3072, 32, 32)
    feature: outputs, the three weight gradients and the data gradient."""
    rs = np.random.RandomState(17)
    A = 21
    x = rs.standard_normal((B, 3072, 32, 32)).astype(np.float32)
    w1 = (rs.standard_normal((512, 3072, 3, 3)) * np.sqrt(2.0 / (3072 * 9))).astype(np.float32)
    b1 = (rs.standard_normal(512) * 0.1).astype(np.float32)
    w2 = (rs.standard_normal((2 * A, 512, 1, 1)) * np.sqrt(1.0 / 512)).astype(np.float32)
    b2 = (rs.standard_normal(2 * A) * 0.1).astype(np.float32)
    w3 = (rs.standard_normal((4 * A, 512, 1, 1)) * np.sqrt(1.0 / 512)).astype(np.float32)
    b3 = (rs.standard_normal(4 * A) * 0.1).astype(np.float32)
    g2 = rs.standard_normal((B, 2 * A, 32, 32)).astype(np.float32)
    g3 = rs.standard_normal((B, 4 * A, 32, 32)).astype(np.float32)
    # oracle: fp32 everywhere, nothing rounded
    xt = torch.from_numpy(x).requires_grad_(True)
    ps = [torch.from_numpy(a).requires_grad_(True) for a in (w1, b1, w2, b2, w3, b3)]
    # device: fp16 storage
    l1, l2, l3 = _Layer(w1, b1, 3, 1, 1, False), _Layer(w2, b2, 1, 0, 0, True), _Layer(w3, b3, 1, 0, 0, True)
    d_y1 = l1.fwd(to_nhwc_f16(x))
    d_cls, d_bbox = l2.fwd(d_y1), l3.fwd(d_y1)
    torch.cuda.synchronize()
    z1 = Fnn.conv2d(xt, ps[0], ps[1], 1, 1)
    y1 = torch.relu(z1)                                             # compared with the device's output
    y1_tf = z1 * torch.from_numpy((from_nhwc(d_y1) > 0).astype(np.float32))     # the device's ReLU decisions, the oracle's values
    cls, bbox = Fnn.conv2d(y1_tf, ps[2], ps[3]), Fnn.conv2d(y1_tf, ps[4], ps[5])
    (cls * torch.from_numpy(g2)).sum().backward(retain_graph=True)
    (bbox * torch.from_numpy(g3)).sum().backward()
    dx2, dw2 = l2.bwd(_dy16(g2, l2.Op))
    dx23, dw3 = l3.bwd(_dy16(g3, l3.Op), dx_acc=dx2)
    dx1, dw1 = l1.bwd(_relu_bwd(dx23, d_y1))
    torch.cuda.synchronize()
    _check('rpn_conv_3x3 output', from_nhwc(d_y1), y1.detach().numpy())
    _check('rpn_cls_score output', from_nhwc(d_cls), cls.detach().numpy())
    _check('rpn_bbox_pred output', from_nhwc(d_bbox), bbox.detach().numpy())
    _check('rpn_cls_score dW', _dw_oihw(dw2, 1), ps[2].grad.numpy())
    _check('rpn_bbox_pred dW', _dw_oihw(dw3, 1), ps[4].grad.numpy())
    _check('rpn_conv_3x3 dW', _dw_oihw(dw1, 3), ps[0].grad.numpy())
    _check('rpn_conv_3x3 dX', from_nhwc(dx1), xt.grad.numpy())


def test_rcnn_head_against_unrounded_fp32():
    """fc_new_1 (12544 -> 1024, ReLU) -> fc_new_2 (1024 -> 1024, ReLU) -> cls_score (81) / bbox_pred (8) on 6000 RoIs."""
    rs = np.random.RandomState(23)
    R = B * 300
    x = np.maximum(rs.standard_normal((R, 12544, 1, 1)), 0).astype(np.float32)         # pooled features are post-ReLU
    shp = [(1024, 12544), (1024, 1024), (81, 1024), (8, 1024)]
    ws = [(rs.standard_normal(s + (1, 1)) * np.sqrt(2.0 / s[1])).astype(np.float32) for s in shp]
    bs = [(rs.standard_normal(s[0]) * 0.1).astype(np.float32) for s in shp]
    g3 = rs.standard_normal((R, 81, 1, 1)).astype(np.float32)
    g4 = rs.standard_normal((R, 8, 1, 1)).astype(np.float32)
    xt = torch.from_numpy(x).requires_grad_(True)
    wt = [torch.from_numpy(a).requires_grad_(True) for a in ws]
    bt = [torch.from_numpy(a).requires_grad_(True) for a in bs]
    ls = [_Layer(ws[0], bs[0], 1, 0, 1, False), _Layer(ws[1], bs[1], 1, 0, 1, False), _Layer(ws[2], bs[2], 1, 0, 0, True),
          _Layer(ws[3], bs[3], 1, 0, 0, True)]
    d_h1 = ls[0].fwd(to_nhwc_f16(x))
    d_h2 = ls[1].fwd(d_h1)
    d_cls, d_bbox = ls[2].fwd(d_h2), ls[3].fwd(d_h2)
    torch.cuda.synchronize()
    z1 = Fnn.conv2d(xt, wt[0], bt[0])
    h1 = torch.relu(z1)
    h1_tf = z1 * torch.from_numpy((from_nhwc(d_h1) > 0).astype(np.float32))
    z2 = Fnn.conv2d(h1_tf, wt[1], bt[1])
    h2 = torch.relu(z2)
    h2_tf = z2 * torch.from_numpy((from_nhwc(d_h2) > 0).astype(np.float32))
    cls, bbox = Fnn.conv2d(h2_tf, wt[2], bt[2]), Fnn.conv2d(h2_tf, wt[3], bt[3])
    (cls * torch.from_numpy(g3)).sum().backward(retain_graph=True)
    (bbox * torch.from_numpy(g4)).sum().backward()
    dx3, dw3 = ls[2].bwd(_dy16(g3, ls[2].Op))
    dx34, dw4 = ls[3].bwd(_dy16(g4, ls[3].Op), dx_acc=dx3)
    dx2, dw2 = ls[1].bwd(_relu_bwd(dx34, d_h2))
    dx1, dw1 = ls[0].bwd(_relu_bwd(dx2, d_h1))
    torch.cuda.synchronize()
    _check('fc_new_1 output', from_nhwc(d_h1), h1.detach().numpy())
    _check('fc_new_2 output', from_nhwc(d_h2), h2.detach().numpy())
    _check('cls_score output', from_nhwc(d_cls), cls.detach().numpy())
    _check('bbox_pred output', from_nhwc(d_bbox), bbox.detach().numpy())
    _check('cls_score dW', _dw_oihw(dw3, 1), wt[2].grad.numpy())
    _check('bbox_pred dW', _dw_oihw(dw4, 1), wt[3].grad.numpy())
    _check('fc_new_2 dW', _dw_oihw(dw2, 1), wt[1].grad.numpy())
    _check('fc_new_1 dW', _dw_oihw(dw1, 1), wt[0].grad.numpy())
    _check('fc_new_1 dX', from_nhwc(dx1), xt.grad.numpy())
