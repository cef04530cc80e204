"""-m gpu: the hot-path kernels at BASELINE configs[1] FULL sizes (20 chips of 512 x 512, R101 shapes), where the CPU
oracle would take minutes: size-independent properties of the domain instead of element-wise comparison --
linearity of the convolutions, conservation laws of the pooling gradients, normalisation of BatchNorm, idempotence and
order of NMS, coverage of the chip generator."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import dev  # noqa: E402

B = 20


def _hip():
    from sniper_amd import hip
    return hip


def _h(*s, scale=0.5):
    return (torch.randn(*s, device=dev()) * scale).half()


@pytest.mark.parametrize('H,C,O,K,pad,dil', [(32, 3072, 512, 3, 1, 1), (32, 1024, 256, 1, 0, 1), (64, 128, 128, 3, 1, 1),
                                             (32, 512, 512, 3, 2, 2)])
def test_conv_linearity_and_adjointness_full_size(H, C, O, K, pad, dil):
    """conv is linear in x; dgrad is its adjoint (<conv(x), dy> == <x, dgrad(dy)>); wgrad is the adjoint in w
    (<conv_w(x), dy> == <w, wgrad(dy, x)>).  Inner products in float64 on the host side of fp16 tensors."""
    hip = _hip()
    torch.manual_seed(H + C + K)
    x1, x2, w = _h(B, H, H, C), _h(B, H, H, C), _h(O, K * K, C, scale=0.05)
    y = lambda x: (lambda out: (hip.call('sn_conv_fwd', x, w, None, None, out, B, H, H, C, C, O, O, 0, K, K, 1, pad, dil, 0, 1,
                                         hip.stream()), out)[1])(torch.empty((B, H, H, O), dtype=torch.float32, device=dev()))
    y1, y2 = y(x1), y(x2)
    y12 = y((x1.float() * 0.5 + x2.float() * 0.25).half())
    want = 0.5 * y1 + 0.25 * y2
    rel = float((y12 - want).norm() / want.norm())
    assert rel < 2e-3, rel                       # fp16 rounding of the combined input only
    dy = _h(B, H, H, O, scale=0.1)
    wT = torch.empty((C, K * K, O), dtype=torch.float16, device=dev())
    hip.call('sn_weight_transpose', w.float().contiguous(), wT, O, K * K, C, O, hip.stream())
    dx = torch.empty((B, H, H, C), dtype=torch.float32, device=dev())
    hip.call('sn_conv_dgrad', dy, wT, None, dx, B, H, H, C, C, O, O, C, K, K, 1, pad, dil, 1, hip.stream())
    lhs = float((y1.double() * dy.double()).sum())
    rhs = float((x1.double() * dx.double()).sum())
    assert abs(lhs - rhs) <= 2e-3 * max(abs(lhs), abs(rhs), 1.0), (lhs, rhs)
    dw = torch.zeros((O, K * K, C), dtype=torch.float32, device=dev())
    need = hip.query('sn_conv_wgrad_workspace_bytes', B, H, H, C, C, O, O, K, K, 1, pad, dil)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev())
    hip.call('sn_conv_wgrad', dy, x1, dw, B, H, H, C, C, O, O, K, K, 1, pad, dil, ws, need, hip.stream())
    rhs_w = float((w.double() * dw.double()).sum())
    assert abs(lhs - rhs_w) <= 2e-3 * max(abs(lhs), abs(rhs_w), 1.0), (lhs, rhs_w)
    # deterministic: the split-K reduction has a fixed order
    dw2 = torch.zeros_like(dw)
    hip.call('sn_conv_wgrad', dy, x1, dw2, B, H, H, C, C, O, O, K, K, 1, pad, dil, ws, need, hip.stream())
    assert torch.equal(dw, dw2)


def test_batchnorm_normalises_full_size():
    hip = _hip()
    M, C = B * 64 * 64, 512
    x = (torch.randn(M, C, device=dev()) * 3 + 1.5).half()
    ws = torch.empty(hip.query('sn_bn_workspace_bytes', M, C), dtype=torch.uint8, device=dev())
    f = lambda: torch.zeros(C, device=dev())
    g, b, rm, rv, sc, sh, mu, iv = torch.ones(C, device=dev()), f(), f(), f(), f(), f(), f(), f()
    hip.call('sn_bn_stats', x, M, C, C, ws, hip.stream())
    hip.call('sn_bn_finalize', ws, M, C, 2e-5, 0.9, g, b, rm, rv, sc, sh, mu, iv, hip.stream())
    y = torch.empty_like(x)
    hip.call('sn_bn_apply', x, y, M, C, C, C, sc, sh, 0, hip.stream())
    yf = y.float()
    assert float(yf.mean(0).abs().max()) < 2e-3 and float((yf.var(0, unbiased=False) - 1).abs().max()) < 5e-3
    assert float((mu - x.float().mean(0)).abs().max()) < 1e-3
    # backward: the gradient w.r.t. x of a normalised output is orthogonal to 1 and to xhat (per channel)
    dy = _h(M, C, scale=1.0)
    dx = torch.empty_like(x)
    dg, db = f(), f()
    hip.call('sn_bn_backward', dy, x, None, dx, M, C, C, C, C, C, sc, sh, mu, iv, 0, ws, dg, db, hip.stream())
    dxf = dx.float()
    xhat = (x.float() - mu) * iv
    scale_ref = float(dy.float().abs().sum(0).max())
    assert float(dxf.sum(0).abs().max()) < 2e-3 * scale_ref
    assert float((dxf * xhat).sum(0).abs().max()) < 2e-3 * scale_ref
    assert float((db - dy.float().sum(0)).abs().max()) < 1e-3 * scale_ref


def test_dpsroi_conservation_full_size():
    """R = 6000 RoIs on 20 maps of 32 x 32 x 256: a constant map pools to the constant; the data gradient conserves
    mass (bilinear weights of a sample sum to one): sum_cells d_data[c] == sum over bins with a valid sample of dout[c]."""
    hip = _hip()
    rs = np.random.RandomState(0)
    R, C, P, S = B * 300, 256, 7, 4
    rois = np.zeros((R, 5), np.float32)
    rois[:, 0] = np.repeat(np.arange(B), 300)
    c = rs.uniform(0, 512, (R, 2))
    wh = np.exp(rs.uniform(np.log(8), np.log(400), (R, 2)))
    rois[:, 1:3], rois[:, 3:5] = np.clip(c - wh / 2, 0, 511), np.clip(c + wh / 2, 0, 511)
    d_rois = torch.from_numpy(rois).to(dev())
    const = torch.full((B, 32, 32, C), 0.75, dtype=torch.float16, device=dev())
    out = torch.empty((R, P, P, C), dtype=torch.float16, device=dev())
    hip.call('sn_dpsroi_pool_fwd', const, d_rois, None, out, R, 32, 32, C, P, S, 1 / 16., 0.0, hip.stream())
    o = out.float()
    assert float((o - 0.75).abs().max()) < 2e-3          # every bin of an in-image RoI has at least one valid sample
    dout = _h(R, P, P, C, scale=1.0)
    dd = torch.empty((B, 32, 32, C), dtype=torch.float32, device=dev())
    ws = torch.empty(hip.query('sn_dpsroi_bwd_workspace_bytes', R), dtype=torch.uint8, device=dev())
    hip.call('sn_dpsroi_pool_bwd', dout, const, d_rois, None, dd, 1, None, R, B, 32, 32, C, P, S, 1 / 16., 0.0, ws, hip.stream())
    got = dd.double().sum((1, 2))                                             # (B, C)
    want = dout.double().view(B, 300 * P * P, C).sum(1)
    assert float((got - want).abs().max()) <= 1e-3 * float(want.abs().max()) + 1e-2, float((got - want).abs().max())


def test_nms_idempotent_and_sorted_full_size():
    from sniper_amd.ext import gpu_nms
    rs = np.random.RandomState(1)
    N = 6000
    c = rs.uniform(0, 512, (B, N, 2))
    wh = np.exp(rs.uniform(np.log(8), np.log(300), (B, N, 2)))
    sc = np.sort(rs.uniform(0, 1, (B, N, 1)), 1)[:, ::-1]
    dets = np.concatenate((c - wh / 2, c + wh / 2, sc), 2).astype(np.float32)
    d = torch.from_numpy(dets).to(dev())
    keep, nk = gpu_nms.nms_sorted_device(d, 0.7, 0)
    keep, nk = keep.cpu().numpy(), nk.cpu().numpy()
    for b in range(B):
        k = keep[b, :nk[b]]
        assert (np.diff(k) > 0).all()                        # survivors come out in score order
        surv = dets[b, k]
        d2 = torch.from_numpy(np.ascontiguousarray(surv[None])).to(dev())
        keep2, nk2 = gpu_nms.nms_sorted_device(d2, 0.7, 0)
        assert int(nk2[0]) == len(k)                          # idempotent: survivors do not suppress each other
        # every suppressed box overlaps a better survivor by more than the threshold
        sup = np.setdiff1d(np.arange(N), k)[:50]
        for i in sup:
            bx = dets[b, i]
            better = surv[k < i]
            iw = np.minimum(bx[2], better[:, 2]) - np.maximum(bx[0], better[:, 0]) + 1
            ih = np.minimum(bx[3], better[:, 3]) - np.maximum(bx[1], better[:, 1]) + 1
            inter = np.maximum(iw, 0) * np.maximum(ih, 0)
            ua = (bx[2] - bx[0] + 1) * (bx[3] - bx[1] + 1) + (better[:, 2] - better[:, 0] + 1) * (better[:, 3] - better[:, 1] + 1) - inter
            assert (inter / ua).max() > 0.7 - 1e-6


def test_chip_generation_covers_every_box_full_size():
    """chips::cgenerate is a greedy set cover: every box that fits in some candidate chip ends up inside a selected
    chip (cchips.cpp:131-170), for a whole synthetic roidb in one launch."""
    from sniper_amd.ext import chips as chipmod
    rs = np.random.RandomState(2)
    units = []
    for _ in range(400):
        W, H = int(rs.randint(600, 2000)), int(rs.randint(600, 1500))
        n = int(rs.randint(1, 60))
        c = rs.uniform(0, [W, H], (n, 2))
        wh = rs.uniform(4, 180, (n, 2))
        bx = np.concatenate((np.clip(c - wh / 2, 0, [W - 2, H - 2]), np.clip(c + wh / 2, 0, [W - 2, H - 2])), 1).astype(np.float32)
        units.append((bx, W, H, 512, 56))
    got = chipmod.generate_batch(units)
    for (bx, W, H, _, _), ch in zip(units, got):
        assert 1 <= len(ch) <= len(bx)
        inside = (bx[:, None, 0] >= ch[None, :, 0]) & (bx[:, None, 1] >= ch[None, :, 1]) & (bx[:, None, 2] <= ch[None, :, 2]) & \
            (bx[:, None, 3] <= ch[None, :, 3])
        assert inside.any(1).all()
