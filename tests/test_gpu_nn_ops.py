"""-m gpu: network kernels through the C ABI against fp32 CPU references (torch-CPU for the standard
ops, oracle/nn.py for the fork-resident ones).  Tolerance: fp16 storage / fp32 accumulate -> 1e-2
relative (north_star), tighter where the op is fp32."""
import numpy as np
import pytest
import torch
import torch.nn.functional as Fnn

pytestmark = pytest.mark.gpu

from gpu_util import assert_close, dev, f16r, from_nhwc, to_nhwc_f16, w_to_otI  # noqa: E402
from oracle import nn as onn  # noqa: E402


def _hip():
    from sniper_amd import hip
    return hip


def _conv_fwd(x_nchw, w_oihw, bias=None, res_nchw=None, stride=1, pad=0, dil=1, relu=0, out_f32=0):
    hip = _hip()
    N, C, H, W = x_nchw.shape
    O, I, KH, KW = w_oihw.shape
    Ho = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
    x = to_nhwc_f16(x_nchw)
    w = torch.from_numpy(w_to_otI(w_oihw)).to(dev()).half().contiguous()
    y = torch.empty((N, Ho, Wo, O), dtype=torch.float32 if out_f32 else torch.float16, device=dev())
    b = torch.from_numpy(bias.astype(np.float32)).to(dev()) if bias is not None else None
    r = to_nhwc_f16(res_nchw) if res_nchw is not None else None
    hip.call('sn_conv_fwd', x, w, b, r, y, N, H, W, C, C, O, O, O, KH, KW, stride, pad, dil, relu, out_f32, hip.stream())
    torch.cuda.synchronize()
    return from_nhwc(y)


def _ref_conv(x, w, bias, res, stride, pad, dil, relu):
    y = Fnn.conv2d(torch.from_numpy(f16r(x)), torch.from_numpy(f16r(w)),
                   None if bias is None else torch.from_numpy(bias.astype(np.float32)), stride, pad, dil).numpy()
    if res is not None:
        y = y + f16r(res)
    if relu:
        y = np.maximum(y, 0)
    return y


def test_conv_mfma_layout_probe():
    """Tiny 1x1 conv with a permutation-like weight: a wrong MFMA fragment / C-D layout assumption
    shows up as a recognisable permutation instead of noise (asymmetric operands, cdna guide G9)."""
    rs = np.random.RandomState(0)
    x = rs.randint(-8, 9, size=(1, 32, 4, 8)).astype(np.float32)         # M = 32 pixels, K = 32
    w = np.zeros((16, 32, 1, 1), np.float32)
    for o in range(16):
        w[o, (3 * o + 1) % 32, 0, 0] = 1.0 + o                               # asymmetric
    got = _conv_fwd(x, w, out_f32=1)
    want = _ref_conv(x, w, None, None, 1, 0, 1, 0)
    assert_close(got, want, 0, 1e-3, 'mfma layout probe')


CONV_CASES = [
    # N, C, H, W, O, K, stride, pad, dil, bias, res, relu
    (2, 64, 16, 16, 128, 3, 1, 1, 1, False, False, 0),
    (2, 64, 16, 16, 64, 1, 1, 0, 1, False, True, 0),
    (1, 128, 17, 13, 256, 3, 2, 1, 1, False, False, 0),
    (2, 256, 8, 8, 128, 3, 1, 2, 2, False, False, 1),
    (3, 512, 8, 8, 42, 1, 1, 0, 1, True, False, 0),
    (2, 24, 10, 10, 72, 3, 1, 1, 1, True, False, 1),     # Cin not a multiple of 32
    (1, 256, 32, 32, 128, 1, 2, 0, 1, False, False, 0),  # stride-2 1x1 shortcut
    (2, 8, 9, 9, 16, 3, 1, 1, 1, False, False, 0),
    (2, 320, 20, 24, 136, 3, 1, 1, 1, False, False, 0),  # partial 128-tiles in both channel dims, Wo < 32
    (2, 256, 32, 32, 256, 3, 1, 1, 1, False, False, 0),  # several K-splits of the weight gradient
    (1, 128, 40, 72, 64, 3, 2, 1, 1, False, False, 0),   # Wo = 36: two 32-pixel units per row, stride 2
    (2, 128, 12, 12, 256, 1, 1, 0, 1, True, True, 1),    # pipelined kernel: bias + residual + ReLU epilogue
    (1, 192, 9, 9, 130, 3, 1, 1, 1, True, False, 0),     # pipelined kernel, Cout % 4 != 0: scalar epilogue, 3 K-steps/tap
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fwd(case):
    N, C, H, W, O, K, s, p, d, hb, hr, relu = case
    rs = np.random.RandomState(hash(case) % 2**31)
    x = rs.standard_normal((N, C, H, W)).astype(np.float32)
    w = (rs.standard_normal((O, C, K, K)) / np.sqrt(C * K * K)).astype(np.float32)
    b = rs.standard_normal(O).astype(np.float32) if hb else None
    want0 = _ref_conv(x, w, b, None, s, p, d, 0)
    res = rs.standard_normal(want0.shape).astype(np.float32) if hr else None
    want = _ref_conv(x, w, b, res, s, p, d, relu)
    for out_f32 in (0, 1):
        got = _conv_fwd(x, w, b, res, s, p, d, relu, out_f32)
        assert_close(got, want, 1e-2, 1e-2 * np.abs(want).max(), 'conv fwd %s f32=%d' % (case, out_f32))


def test_conv_stem_packed_7x7():
    """conv0 (resnet_mx_101_e2e.py:402-404): bn_data affine + 7x7/2 pad 3 on the packed NHWC4 input."""
    hip = _hip()
    rs = np.random.RandomState(1)
    N, H, W, O = 2, 64, 64, 64
    x = (rs.standard_normal((N, 3, H, W)) * 50).astype(np.float32)
    w = (rs.standard_normal((O, 3, 7, 7)) / np.sqrt(147)).astype(np.float32)
    scale, shift = rs.uniform(0.5, 1.5, 3).astype(np.float32), rs.standard_normal(3).astype(np.float32)
    Hp, Wp = H + 6, W + 8
    xp = torch.empty((N, Hp, Wp, 4), dtype=torch.float16, device=dev())
    hip.call('sn_pack_stem_input', torch.from_numpy(x).to(dev()), xp, N, 3, H, W, Hp, Wp, 3, 3,
             torch.from_numpy(scale).to(dev()), torch.from_numpy(shift).to(dev()), hip.stream())
    wk = np.zeros((O, 7, 8, 4), np.float32)
    wk[:, :, :7, :3] = w.transpose(0, 2, 3, 1)                       # [o][kh][kw][ci], kw=7 / ci=3 zero
    wd = torch.from_numpy(wk.reshape(O, 7, 32)).to(dev()).half().contiguous()
    Ho, Wo = H // 2, W // 2
    y = torch.empty((N, Ho, Wo, O), dtype=torch.float16, device=dev())
    hip.call('sn_conv_stem_fwd', xp, wd, None, y, N, Hp, Wp, Ho, Wo, O, O, 7, 8, 2, 0, 0, hip.stream())
    torch.cuda.synchronize()
    got = from_nhwc(y)
    xa = x * scale[None, :, None, None] + shift[None, :, None, None]
    want = Fnn.conv2d(torch.from_numpy(f16r(xa)), torch.from_numpy(f16r(w)), None, 2, 3).numpy()
    assert_close(got, want, 1e-2, 1e-2 * np.abs(want).max(), 'stem conv')


@pytest.fixture
def conv_tuning():
    """Restores the built-in kernel selection after a test that forced one (sn_conv_tune / sn_conv_wgrad_impl are process-wide)."""
    hip = _hip()
    yield hip
    hip.call('sn_conv_tune', -1)
    hip.call('sn_conv_wgrad_impl', 1, 0)


@pytest.mark.parametrize('case', CONV_CASES[:7] + CONV_CASES[8:])
def test_conv_dgrad_wgrad(case):
    _check_dgrad_wgrad(case)


# layers the pipelined kernels accept (Cin % 64 == 0, Cout > 64): every LDS-DMA tile configuration, forward + data gradient
DMA_CASES = [CONV_CASES[i] for i in (0, 2, 3, 8, 9, 11, 12)]


@pytest.mark.parametrize('cfg', [4, 5, 6, 7, 14, 16, 18])
def test_conv_dma_configurations(cfg, conv_tuning):
    """conv_dma_kernel (csrc/conv_dma.hip): each (tile, waves, ring depth) configuration forced on layers with padding taps,
    stride 2, dilation, ragged M / Cout tiles, bias + residual + ReLU and the scalar epilogue, against torch-CPU fp32."""
    conv_tuning.call('sn_conv_tune', cfg)
    for case in DMA_CASES:
        N, C, H, W, O, K, s, p, d, hb, hr, relu = case
        rs = np.random.RandomState(cfg * 131 + hash(case) % 2**20)
        x = rs.standard_normal((N, C, H, W)).astype(np.float32)
        w = (rs.standard_normal((O, C, K, K)) / np.sqrt(C * K * K)).astype(np.float32)
        b = rs.standard_normal(O).astype(np.float32) if hb else None
        want0 = _ref_conv(x, w, b, None, s, p, d, 0)
        res = rs.standard_normal(want0.shape).astype(np.float32) if hr else None
        want = _ref_conv(x, w, b, res, s, p, d, relu)
        got = _conv_fwd(x, w, b, res, s, p, d, relu, 0)
        assert_close(got, want, 1e-2, 1e-2 * np.abs(want).max(), 'conv fwd cfg %d %s' % (cfg, case))
        _check_dgrad_wgrad(case, wgrad=False)


@pytest.mark.parametrize('mode', [(1, 0), (1, 2), (1, 7), (1, 1000), (0, 0)])
def test_conv_wgrad_kernels(mode, conv_tuning):
    """wgrad_ps_kernel (csrc/conv_wgrad_ps.hip: 4 consumer + 4 producer waves, jobs of a batched table) forced to job lengths of
    2 / 7 / 1000 K-steps -- many K-splits with slabs + reduce, odd unit counts, no split at all -- and the gather kernel every
    layer can fall back to (impl 0), on 1x1 and 3x3 layers (stride 2, dilation 2, Wo < 32 and Wo = 32, ragged channel tiles,
    Cout % 8 != 0), against torch-CPU fp32."""
    conv_tuning.call('sn_conv_wgrad_impl', *mode)
    for i in (0, 1, 3, 4, 8, 9, 11, 12):
        _check_dgrad_wgrad(CONV_CASES[i], dgrad=False)


def test_conv_wgrad_batch_equals_per_layer_launches(conv_tuning):
    """sn_conv_wgrad_batch: a table of different layers in one launch (whole-K jobs, some problems split) gives each layer the
    result of its own sn_conv_wgrad launch up to fp32 summation order, and adds into non-zero dw (+= semantics)."""
    hip = _hip()
    rs = np.random.RandomState(5)
    probs, singles = [], []
    for (N, C, H, W, O, K, s, p, d) in [(3, 128, 20, 20, 256, 1, 1, 0, 1), (3, 64, 20, 20, 72, 3, 1, 2, 2), (2, 256, 40, 33, 128, 3, 2, 1, 1),
                                        (600, 192, 1, 1, 81, 1, 1, 0, 1), (3, 128, 20, 20, 128, 3, 1, 1, 1)]:
        Ho, Wo = (H + 2 * p - d * (K - 1) - 1) // s + 1, (W + 2 * p - d * (K - 1) - 1) // s + 1
        Op = (O + 7) // 8 * 8
        x = torch.from_numpy(rs.standard_normal((N, H, W, C)).astype(np.float32)).to(dev()).half()
        dy = torch.zeros((N, Ho, Wo, Op), dtype=torch.float16, device=dev())
        dy[..., :O] = torch.from_numpy(rs.standard_normal((N, Ho, Wo, O)).astype(np.float32)).to(dev()).half()
        init = torch.from_numpy(rs.standard_normal((O, K * K, C)).astype(np.float32)).to(dev())
        probs.append((dy, x, init.clone(), N, H, W, C, C, O, Op, K, K, s, p, d))
        one = init.clone()
        need = hip.query('sn_conv_wgrad_workspace_bytes', N, H, W, C, C, O, Op, K, K, s, p, d)
        ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev())
        hip.call('sn_conv_wgrad', dy, x, one, N, H, W, C, C, O, Op, K, K, s, p, d, ws, need, hip.stream())
        singles.append((one, init))
    tab = hip.wgrad_table(probs)
    need = hip.query('sn_conv_wgrad_batch_workspace_bytes', tab, len(probs))
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev())
    hip.call('sn_conv_wgrad_batch', tab, len(probs), ws, need, hip.stream())
    torch.cuda.synchronize()
    for pr, (one, init) in zip(probs, singles):
        got, want = (pr[2] - init).cpu().numpy(), (one - init).cpu().numpy()
        assert np.abs(want).max() > 1.0
        assert_close(got, want, 0.0, 2e-5 * np.abs(want).max(), 'batched wgrad %s' % (pr[3:],))


def _check_dgrad_wgrad(case, dgrad=True, wgrad=True):
    hip = _hip()
    N, C, H, W, O, K, s, p, d, _, _, _ = case
    rs = np.random.RandomState(7 + hash(case) % 1000)
    x = rs.standard_normal((N, C, H, W)).astype(np.float32)
    w = (rs.standard_normal((O, C, K, K)) / np.sqrt(C * K * K)).astype(np.float32)
    xt = torch.from_numpy(f16r(x)).requires_grad_(True)
    wt = torch.from_numpy(f16r(w)).requires_grad_(True)
    y = Fnn.conv2d(xt, wt, None, s, p, d)
    dy = rs.standard_normal(tuple(y.shape)).astype(np.float32)
    y.backward(torch.from_numpy(f16r(dy)))
    want_dx, want_dw = xt.grad.numpy(), wt.grad.numpy()
    Ho, Wo = y.shape[2], y.shape[3]
    # the contraction length of dgrad (Cout) must be 8-aligned: dY and W^T are zero padded (RPN heads: 42 -> 48)
    Op = (O + 7) // 8 * 8
    d_dy = torch.zeros((N, Ho, Wo, Op), dtype=torch.float16, device=dev())
    d_dy[..., :O] = to_nhwc_f16(dy)
    # dgrad: weights as [Cin][taps][Cout]
    w_otI = torch.from_numpy(w_to_otI(w)).to(dev())
    wT = torch.empty((C, K * K, Op), dtype=torch.float16, device=dev())
    hip.call('sn_weight_transpose', w_otI, wT, O, K * K, C, Op, hip.stream())
    if dgrad:
        dx = torch.empty((N, H, W, C), dtype=torch.float16, device=dev())
        hip.call('sn_conv_dgrad', d_dy, wT, None, dx, N, H, W, C, C, Op, Op, C, K, K, s, p, d, 0, hip.stream())
        torch.cuda.synchronize()
        assert_close(from_nhwc(dx), want_dx, 1e-2, 1e-2 * np.abs(want_dx).max(), 'dgrad %s' % (case,))
        # accumulate form: dx2 = dgrad + dx
        dx2 = dx.clone()
        hip.call('sn_conv_dgrad', d_dy, wT, dx2, dx2, N, H, W, C, C, Op, Op, C, K, K, s, p, d, 0, hip.stream())
        torch.cuda.synchronize()
        assert_close(from_nhwc(dx2), 2 * want_dx, 2e-2, 2e-2 * np.abs(want_dx).max(), 'dgrad accumulate')
    if not wgrad:
        return
    # wgrad (+= into zeroed fp32)
    dw = torch.zeros((O, K * K, C), dtype=torch.float32, device=dev())
    need = hip.query('sn_conv_wgrad_workspace_bytes', N, H, W, C, C, O, Op, K, K, s, p, d)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev())
    hip.call('sn_conv_wgrad', d_dy, to_nhwc_f16(x), dw, N, H, W, C, C, O, Op, K, K, s, p, d, ws, need, hip.stream())
    torch.cuda.synchronize()
    got_dw = dw.cpu().numpy().reshape(O, K, K, C).transpose(0, 3, 1, 2)
    assert_close(got_dw, want_dw, 1e-2, 1e-2 * np.abs(want_dw).max(), 'wgrad %s' % (case,))
    # without scratch the layer runs unsplit (one owner per element, no atomics): same result, and += semantics on a non-zero dw
    dw2 = dw.clone()
    hip.call('sn_conv_wgrad', d_dy, to_nhwc_f16(x), dw2, N, H, W, C, C, O, Op, K, K, s, p, d, None, 0, hip.stream())
    torch.cuda.synchronize()
    got2 = dw2.cpu().numpy().reshape(O, K, K, C).transpose(0, 3, 1, 2)
    assert_close(got2, 2 * want_dw, 1e-2, 2e-2 * np.abs(want_dw).max(), 'wgrad without scratch %s' % (case,))


def test_fc_as_conv_and_bias_grad():
    hip = _hip()
    rs = np.random.RandomState(2)
    M, K, O = 300, 12544, 98
    x = rs.standard_normal((M, K)).astype(np.float32)
    w = (rs.standard_normal((O, K)) / np.sqrt(K)).astype(np.float32)
    b = rs.standard_normal(O).astype(np.float32)
    xd = torch.from_numpy(x).to(dev()).half()
    wd = torch.from_numpy(w).to(dev()).half()
    y = torch.empty((M, O), dtype=torch.float32, device=dev())
    hip.call('sn_conv_fwd', xd, wd, torch.from_numpy(b).to(dev()), None, y, M, 1, 1, K, K, O, O, O, 1, 1, 1, 0, 1, 0, 1, hip.stream())
    want = f16r(x) @ f16r(w).T + b
    assert_close(y.cpu().numpy(), want, 1e-2, 1e-2 * np.abs(want).max(), 'fc fwd')
    dy = rs.standard_normal((M, O)).astype(np.float32)
    dyd = torch.from_numpy(dy).to(dev())
    need = hip.query('sn_bias_grad_workspace_bytes', M, O)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev())
    runs = []
    for rep in range(3):         # ordered partial sums: bit-identical every run, with or without scratch of its own order
        db = torch.zeros(O, dtype=torch.float32, device=dev())
        hip.call('sn_bias_grad', dyd, db, M, O, O, 1, ws, need, hip.stream())
        runs.append(db.clone())
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
    assert_close(db.cpu().numpy(), dy.sum(0), 1e-4, 1e-3, 'bias grad')
    db1 = torch.zeros(O, dtype=torch.float32, device=dev())
    hip.call('sn_bias_grad', dyd, db1, M, O, O, 1, None, 0, hip.stream())
    assert_close(db1.cpu().numpy(), dy.sum(0), 1e-4, 1e-3, 'bias grad without scratch')
    # wgrad of an FC needs fp16 dy with an 8-aligned row stride: pad 98 -> 104
    dy16 = torch.zeros((M, 104), dtype=torch.float16, device=dev())
    dy16[:, :O] = dyd.half()
    dw = torch.zeros((O, K), dtype=torch.float32, device=dev())
    hip.call('sn_conv_wgrad', dy16, xd, dw, M, 1, 1, K, K, O, 104, 1, 1, 1, 0, 1, None, 0, hip.stream())
    want_dw = f16r(dy).T @ f16r(x)
    assert_close(dw.cpu().numpy(), want_dw, 1e-2, 1e-2 * np.abs(want_dw).max(), 'fc wgrad')


@pytest.mark.parametrize('rows,C,ld', [(20480, 512, 512), (6000, 1024, 1024), (20480, 72, 72), (777, 256, 264), (20480, 42, 48), (6000, 81, 88),
                                       (20480, 84, 88), (300, 30, 33), (5, 8, 8)])
def test_bias_grad_fp16_paths(rows, C, ld):
    """sn_bias_grad on fp16 gradients (the step's twelve bias gradients): the 16-byte-per-lane form (C and pitch multiples of 8)
    and the 2-byte form, with and without scratch, against a float64 column sum; ordered partial sums -> the same bits every run;
    `+=` semantics on a non-zero db."""
    hip = _hip()
    rs = np.random.RandomState(rows + C)
    dy = rs.standard_normal((rows, ld)).astype(np.float32)
    dyd = torch.from_numpy(dy).to(dev()).half()
    want = dyd[:, :C].double().sum(0).cpu().numpy()
    need = hip.query('sn_bias_grad_workspace_bytes', rows, C)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev())
    runs = []
    for rep in range(2):
        db = torch.full((C,), 1.5, dtype=torch.float32, device=dev())
        hip.call('sn_bias_grad', dyd, db, rows, C, ld, 0, ws, need, hip.stream())
        runs.append(db.clone())
    assert torch.equal(runs[0], runs[1])
    assert_close(runs[0].cpu().numpy() - 1.5, want, 1e-4, 1e-3 * max(1.0, np.sqrt(rows) / 10), 'bias grad fp16')
    db1 = torch.zeros(C, dtype=torch.float32, device=dev())
    hip.call('sn_bias_grad', dyd, db1, rows, C, ld, 0, None, 0, hip.stream())
    assert_close(db1.cpu().numpy(), want, 1e-4, 1e-3 * max(1.0, np.sqrt(rows) / 10), 'bias grad fp16 without scratch')


def test_batchnorm_train_fwd_bwd():
    hip = _hip()
    rs = np.random.RandomState(3)
    # C/8 = 32, 8, 256 (one slab), 4, 3, 12 (MobileNetV2 widths: not divisors of 256), 320 (two slabs), many rows
    for (N, C, H, W, relu) in ((4, 256, 16, 16, 1), (2, 64, 9, 7, 0), (3, 2048, 4, 4, 1), (2, 32, 5, 6, 1), (2, 24, 7, 3, 0),
                               (1, 96, 11, 5, 1), (2, 2560, 3, 3, 1), (8, 128, 40, 40, 1)):
        x = (rs.standard_normal((N, C, H, W)) * 2 + 0.5).astype(np.float32)
        gamma, beta = rs.uniform(0.5, 1.5, C).astype(np.float32), rs.standard_normal(C).astype(np.float32) * 0.1
        dy = rs.standard_normal((N, C, H, W)).astype(np.float32)
        M = N * H * W
        xd, dyd = to_nhwc_f16(x), to_nhwc_f16(dy)
        f = lambda n, dt=torch.float32: torch.zeros(n, dtype=dt, device=dev())
        ws = torch.empty(hip.query('sn_bn_workspace_bytes', M, C), dtype=torch.uint8, device=dev())
        scale, shift, mean, invstd = f(C), f(C), f(C), f(C)
        rm, rv = f(C), torch.ones(C, device=dev())
        g_d, b_d = torch.from_numpy(gamma).to(dev()), torch.from_numpy(beta).to(dev())
        hip.call('sn_bn_stats', xd, M, C, C, ws, hip.stream())
        hip.call('sn_bn_finalize', ws, M, C, 2e-5, 0.9, g_d, b_d, rm, rv, scale, shift, mean, invstd, hip.stream())
        y = torch.empty_like(xd)
        hip.call('sn_bn_apply', xd, y, M, C, C, C, scale, shift, relu, hip.stream())
        xt = torch.from_numpy(f16r(x)).requires_grad_(True)
        gt, bt = torch.from_numpy(gamma).requires_grad_(True), torch.from_numpy(beta).requires_grad_(True)
        yt = Fnn.batch_norm(xt, None, None, gt, bt, True, 0.0, 2e-5)
        if relu:
            yt = torch.relu(yt)
        yt.backward(torch.from_numpy(f16r(dy)))
        assert_close(from_nhwc(y), yt.detach().numpy(), 1e-2, 1e-2 * float(yt.detach().abs().max()), 'bn fwd C=%d' % C)
        xm = f16r(x).transpose(1, 0, 2, 3).reshape(C, -1)
        assert_close(rm.cpu().numpy(), 0.1 * xm.mean(1), 1e-3, 1e-4, 'running mean')
        assert_close(rv.cpu().numpy(), 0.9 + 0.1 * xm.var(1), 1e-3, 1e-4, 'running var')
        dx, dg, db = torch.empty_like(xd), f(C), f(C)
        hip.call('sn_bn_backward', dyd, xd, None, dx, M, C, C, C, C, C, scale, shift, mean, invstd, relu, ws, dg, db, hip.stream())
        torch.cuda.synchronize()
        assert_close(dg.cpu().numpy(), gt.grad.numpy(), 1e-2, 1e-2 * np.abs(gt.grad.numpy()).max(), 'dgamma')
        assert_close(db.cpu().numpy(), bt.grad.numpy(), 1e-2, 1e-2 * np.abs(bt.grad.numpy()).max(), 'dbeta')
        assert_close(from_nhwc(dx), xt.grad.numpy(), 1e-2, 1e-2 * np.abs(xt.grad.numpy()).max(), 'bn dx')


@pytest.mark.parametrize('M,C,nblk,relu', [(20480, 256, 128, 1), (20480, 1024, 128, 1), (2048, 64, 13, 0), (5000, 192, 160, 2),
                                           (20480, 256, 512, 1), (3000, 72, 19, 1)])
def test_bn_finalize_fused_into_apply_and_dx_equals_the_separate_launches(M, C, nblk, relu):
    """sn_bn_apply_blocks / sn_bn_backward_blocks (round 6: the finalize launch folded into the consumer, which re-reduces the
    partials of its own 64-channel slab in the separate kernel's summation order) against the separate launches
    (the default; the fused form is opt-in, sn_debug_option bn_fused_finalize: in the step it measured slower): every output BIT-equal -- y, scale, shift, saved statistics, moving averages; dx, dgamma,
    dbeta (accumulated onto a non-zero arena).  nblk = 512 and C = 72 are shapes that keep the separate launches either way."""
    hip = _hip()
    rs = np.random.RandomState(M + C + nblk)
    td = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dev()).to(dt)
    x = td(rs.standard_normal((M, C)) * 1.5 + 0.3, torch.float16)
    dy = td(rs.standard_normal((M, C)), torch.float16)
    acc = td(rs.standard_normal((M, C)), torch.float16)
    xf = x.float()
    # partials as a producer would leave them: per row block sums of x and x^2 (any split of the rows into nblk blocks)
    edges = np.linspace(0, M, nblk + 1).astype(int)
    part = torch.stack([torch.stack((xf[a:b].sum(0), (xf[a:b] ** 2).sum(0))) for a, b in zip(edges[:-1], edges[1:])]).contiguous()
    bpart = td(rs.standard_normal((nblk, 2, C)) * 3)
    gamma, beta = td(rs.uniform(0.5, 1.5, C)), td(rs.standard_normal(C) * 0.1)
    ws = torch.empty(hip.query('sn_bn_workspace_bytes', M, C), dtype=torch.uint8, device=dev())
    res = []
    for fused in (0, 1):
        hip.call('sn_debug_option', b'bn_fused_finalize', fused)
        try:
            f = lambda v=0.0: torch.full((C,), v, dtype=torch.float32, device=dev())
            sc, sh, sm, si, rm, rv = f(7), f(7), f(7), f(7), f(0.25), f(1.5)
            y = torch.full((M, C), 7.0, dtype=torch.float16, device=dev())
            hip.call('sn_bn_apply_blocks', part, nblk, x, y, M, C, C, C, 2e-5, 0.9, gamma, beta, rm, rv, sc, sh, sm, si, relu, hip.stream())
            dx = torch.full((M, C), 7.0, dtype=torch.float16, device=dev())
            dg, db = f(0.5), f(-0.5)
            hip.call('sn_bn_backward_blocks', bpart, nblk, dy, x, acc, dx, M, C, C, C, C, C, sc, sh, sm, si, relu, ws, dg, db, hip.stream())
            dg2, db2 = f(0.0), f(0.0)          # parameter gradients only (dx = NULL: a BatchNorm on a tensor that needs no gradient)
            hip.call('sn_bn_backward_blocks', bpart, nblk, dy, x, None, None, M, C, C, C, C, C, sc, sh, sm, si, relu, ws, dg2, db2, hip.stream())
            torch.cuda.synchronize()
            res.append([t.clone() for t in (y, sc, sh, sm, si, rm, rv, dx, dg, db, dg2, db2)])
        finally:
            hip.call('sn_debug_option', b'bn_fused_finalize', 0)
    names = 'y scale shift save_mean save_invstd run_mean run_var dx dgamma dbeta dgamma_only dbeta_only'.split()
    for n, a, b in zip(names, res[0], res[1]):
        assert torch.equal(a, b), (n, float((a.float() - b.float()).abs().max()))
    # and the statistics are right: mean / variance of the rows
    mean, var = xf.double().mean(0), xf.double().var(0, unbiased=False)
    assert_close(res[1][3].cpu().numpy(), mean.cpu().numpy(), 1e-4, 1e-4, 'saved mean')
    assert_close(res[1][4].cpu().numpy(), (1.0 / torch.sqrt(var + 2e-5)).cpu().numpy(), 1e-4, 1e-4, 'saved invstd')


def test_bn_global_maxpool_ew_layout_ops():
    hip = _hip()
    rs = np.random.RandomState(4)
    N, C, H, W = 2, 64, 18, 14
    x = rs.standard_normal((N, C, H, W)).astype(np.float32)
    g, b, m, v = [rs.uniform(0.5, 1.5, C).astype(np.float32) for _ in range(4)]
    td = lambda a: torch.from_numpy(a).to(dev())
    scale, shift = torch.empty(C, device=dev()), torch.empty(C, device=dev())
    hip.call('sn_bn_global_scale_shift', td(g), td(b), td(m), td(v), C, 2e-5, scale, shift, hip.stream())
    xd = to_nhwc_f16(x)
    y = torch.empty_like(xd)
    hip.call('sn_bn_apply', xd, y, N * H * W, C, C, C, scale, shift, 1, hip.stream())
    want = np.maximum((f16r(x) - m[None, :, None, None]) / np.sqrt(v + 2e-5)[None, :, None, None] * g[None, :, None, None]
                      + b[None, :, None, None], 0)
    assert_close(from_nhwc(y), want, 1e-2, 1e-2, 'bn global')
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    p = torch.empty((N, Ho, Wo, C), dtype=torch.float16, device=dev())
    hip.call('sn_maxpool_fwd', xd, p, N, H, W, C, 3, 2, 1, hip.stream())
    want = Fnn.max_pool2d(torch.from_numpy(f16r(x)), 3, 2, 1).numpy()
    assert_close(from_nhwc(p), want, 0, 0, 'maxpool')
    # ew: relu / add / relu-backward
    a, bb = xd, to_nhwc_f16(rs.standard_normal((N, C, H, W)).astype(np.float32))
    o = torch.empty_like(xd)
    hip.call('sn_ew_f16', a, None, None, o, N * H * W, C, C, C, C, C, 0, hip.stream())
    assert torch.equal(o, torch.relu(a))
    hip.call('sn_ew_f16', a, bb, None, o, N * H * W, C, C, C, C, C, 1, hip.stream())
    assert_close(o.float().cpu().numpy(), (a.float() + bb.float()).cpu().numpy(), 1e-3, 1e-3, 'add')
    hip.call('sn_ew_f16', a, bb, bb, o, N * H * W, C, C, C, C, C, 2, hip.stream())
    want = torch.where(bb > 0, a.float(), torch.zeros_like(a.float())) + bb.float()
    assert_close(o.float().cpu().numpy(), want.cpu().numpy(), 1e-3, 1e-3, 'relu bwd + acc')
    # NHWC(f16) -> NCHW(f32) and back
    nchw = torch.empty((N, C, H, W), dtype=torch.float32, device=dev())
    hip.call('sn_transpose_batched', xd, nchw, N, H * W, C, H * W * C, C * H * W, C, H * W, 0, 1, hip.stream())
    assert_close(nchw.cpu().numpy(), f16r(x), 0, 0, 'nhwc->nchw')
    back = torch.empty_like(xd)
    hip.call('sn_transpose_batched', nchw, back, N, C, H * W, C * H * W, H * W * C, H * W, C, 1, 0, hip.stream())
    assert torch.equal(back, xd)
    # channel-slice copy (Concat): write C channels into a 2C-wide buffer at offset C
    wide = torch.zeros((N, H, W, 2 * C), dtype=torch.float16, device=dev())
    hip.call('sn_copy2d', xd, wide.view(-1)[C:], N * H * W, C, C, 2 * C, 0, 0, hip.stream())
    assert torch.equal(wide[..., C:], xd) and float(wide[..., :C].abs().sum()) == 0


def test_softmax_output_and_smooth_l1():
    hip = _hip()
    rs = np.random.RandomState(5)
    # RPN shape: (B, 2, A*F, F) with labels (B, A*F*F) in {-1, 0, 1}
    B, K, inner = 3, 2, 21 * 8 * 8
    x = rs.standard_normal((B, K, inner)).astype(np.float32)
    lab = rs.choice([-1, 0, 1], size=(B, inner), p=[0.8, 0.15, 0.05]).astype(np.float32)
    xd, ld = torch.from_numpy(x).to(dev()), torch.from_numpy(lab).to(dev())
    p, g = torch.empty_like(xd), torch.empty_like(xd)
    ws = torch.zeros(4, dtype=torch.int32, device=dev())
    hip.call('sn_softmax_fwd', xd, p, B, K, inner, hip.stream())
    hip.call('sn_softmax_output_bwd', p, ld, g, B, K, inner, -1.0, 1, 100.0, 1, ws, hip.stream())
    pt = torch.softmax(torch.from_numpy(x), 1).numpy()
    assert_close(p.cpu().numpy(), pt, 1e-5, 1e-6, 'softmax fwd')
    valid = lab != -1
    onehot = np.zeros_like(x)
    for k in range(K):
        onehot[:, k, :] = (lab == k)
    want = (pt - onehot) * valid[:, None, :] * (100.0 / max(1, valid.sum()))
    assert_close(g.cpu().numpy(), want, 1e-4, 1e-6, 'softmax bwd')
    # RCNN shape: (R, 81) rows
    R, C = 600, 81
    x = rs.standard_normal((R, C)).astype(np.float32)
    lab = rs.randint(-1, C, size=R).astype(np.float32)
    xd, ld = torch.from_numpy(x).to(dev()), torch.from_numpy(lab).to(dev())
    p, g = torch.empty_like(xd), torch.empty_like(xd)
    hip.call('sn_softmax_fwd', xd, p, R, C, 1, hip.stream())
    hip.call('sn_softmax_output_bwd', p, ld, g, R, C, 1, -1.0, 1, 100.0, 1, ws, hip.stream())
    pt = torch.softmax(torch.from_numpy(x), 1).numpy()
    oh = np.zeros_like(x)
    oh[np.arange(R)[lab >= 0], lab[lab >= 0].astype(int)] = 1
    want = (pt - oh) * (lab != -1)[:, None] * (100.0 / (lab != -1).sum())
    assert_close(g.cpu().numpy(), want, 1e-4, 1e-6, 'softmax rcnn bwd')
    # smooth l1
    n = 5000
    a, t, w = [rs.standard_normal(n).astype(np.float32) * 2 for _ in range(3)]
    w = (w > 0).astype(np.float32)
    loss, dp = torch.empty(n, device=dev()), torch.empty(n, device=dev())
    td = lambda z: torch.from_numpy(z).to(dev())
    hip.call('sn_smooth_l1_loss', td(a), td(t), td(w), loss, dp, n, 1.0, 0.25, hip.stream())
    d = a - t
    wl = w * np.where(np.abs(d) < 1, 0.5 * d * d, np.abs(d) - 0.5)
    wg = 0.25 * w * np.where(np.abs(d) < 1, d, np.sign(d))
    assert_close(loss.cpu().numpy(), wl, 1e-5, 1e-6, 'smooth l1')
    assert_close(dp.cpu().numpy(), wg, 1e-5, 1e-6, 'smooth l1 grad')


def test_sgd_and_weight_transpose():
    hip = _hip()
    rs = np.random.RandomState(6)
    n = 100003
    w, g, m = [rs.standard_normal(n).astype(np.float32) for _ in range(3)]
    td = lambda z: torch.from_numpy(z.copy()).to(dev())
    wd, gd, md = td(w), td(g), td(m)
    w16 = torch.empty(n, dtype=torch.float16, device=dev())
    hip.call('sn_sgd_mom_update', wd, gd, md, w16, n, 0.01, 0.001, 0.9, 1.0, hip.stream())
    nm = 0.9 * m - 0.01 * (g + 0.001 * w)
    assert_close(md.cpu().numpy(), nm, 1e-5, 1e-6, 'sgd mom')
    assert_close(wd.cpu().numpy(), w + nm, 1e-5, 1e-6, 'sgd w')
    assert torch.equal(w16, wd.half())
    # the captured-graph flavour (hyper-parameters in device memory) on arena slices that start off a 16-byte boundary:
    # scalar head, four-wide body, scalar tail must equal the plain kernel
    hyper = torch.tensor([0.01, 0.001, 0.9, 0.5], dtype=torch.float32, device=dev())
    for off, cnt in ((0, n), (3, n - 5), (1, 2), (2, 65541), (4, 64)):
        w1, g1, m1, w2, m2 = td(w), td(g), td(m), td(w), td(m)
        h1 = torch.zeros(n, dtype=torch.float16, device=dev())
        h2 = torch.zeros(n, dtype=torch.float16, device=dev())
        hip.call('sn_sgd_mom_update_dev', w1[off:], g1[off:], m1[off:], h1[off:], cnt, hyper, 2.0, 0.5, hip.stream())
        hip.call('sn_sgd_mom_update', w2[off:], g1[off:], m2[off:], h2[off:], cnt, 0.02, 0.0005, 0.9, 0.5, hip.stream())
        # (the two kernels may contract their multiply-adds differently: one ulp, not bit for bit)
        assert torch.allclose(w1, w2, rtol=1e-6, atol=1e-7) and torch.allclose(m1, m2, rtol=1e-6, atol=1e-7), (off, cnt)
        assert torch.equal(h1[off:off + cnt], w1[off:off + cnt].half()) and not h1[:off].any() and not h1[off + cnt:].any()
        assert torch.equal(w1[:off], td(w)[:off]) and torch.equal(w1[off + cnt:], td(w)[off + cnt:])
    O, T, I = 40, 9, 24
    ww = rs.standard_normal((O, T, I)).astype(np.float32)
    wt = torch.empty((I, T, O), dtype=torch.float16, device=dev())
    hip.call('sn_weight_transpose', td(ww), wt, O, T, I, O, hip.stream())
    assert_close(wt.float().cpu().numpy(), f16r(ww.transpose(2, 1, 0)), 0, 0, 'weight transpose')
    wp = torch.empty((I, T, 48), dtype=torch.float16, device=dev())
    hip.call('sn_weight_transpose', td(ww), wp, O, T, I, 48, hip.stream())
    assert torch.equal(wp[..., :O], wt) and float(wp[..., O:].abs().sum()) == 0


CONV_DMA_BM = {4: 128, 5: 64, 6: 64, 7: 256, 14: 160, 16: 160, 18: 160}


@pytest.mark.parametrize('N,H,C,O,K,bm,res', [(3, 17, 64, 192, 1, 'dma5', False), (2, 24, 128, 256, 3, 'dma14', True), (5, 9, 64, 128, 3, 'dma6', True),
                                              (4, 32, 256, 320, 1, 'dma16', False), (3, 17, 64, 192, 1, 'dma6', False),
                                              (4, 32, 256, 320, 1, 'dma4', False), (5, 9, 64, 128, 3, 'dma16', True), (2, 24, 128, 256, 3, 'dma7', True),
                                              (2, 24, 128, 256, 3, 'dma18', True), (4, 32, 256, 320, 1, 'dma18', False)])
def test_conv_fwd_stats_epilogue(N, H, C, O, K, bm, res, monkeypatch, conv_tuning):
    """sn_conv_fwd_stats: same output as sn_conv_fwd, and the per-row-tile partials sum to the statistics of the STORED
    fp16 tensor (what sn_bn_stats would read back); LDS-DMA configurations (4-wave, 8-wave, producer / consumer specialised)
    with 2 / 4 waves along M, ragged M / Cout tiles, residual epilogue."""
    hip = _hip()
    hip.call('sn_conv_tune', int(bm[3:]))
    bm = str(CONV_DMA_BM[int(bm[3:])])
    rs = np.random.RandomState(N * H + O)
    x = rs.standard_normal((N, H, H, C)).astype(np.float32)
    w = (rs.standard_normal((O, K * K, C)) / np.sqrt(K * K * C)).astype(np.float32)
    r = rs.standard_normal((N, H, H, O)).astype(np.float32) if res else None
    xd, wd = torch.from_numpy(x).to(dev()).half(), torch.from_numpy(w).to(dev()).half()
    rd = torch.from_numpy(r).to(dev()).half() if res else None
    y0 = torch.empty((N, H, H, O), dtype=torch.float16, device=dev())
    hip.call('sn_conv_fwd', xd, wd, None, rd, y0, N, H, H, C, C, O, O, O if res else 0, K, K, 1, K // 2, 1, 0, 0, hip.stream())
    nblk = hip.query('sn_conv_fwd_stats_blocks', N, H, H, C, C, O, O, O if res else 0, K, K, 1, K // 2, 1)
    M = N * H * H
    assert nblk == -(-M // int(bm))
    part = torch.full((nblk, 2, O), 7.0, dtype=torch.float32, device=dev())
    y1 = torch.empty_like(y0)
    hip.call('sn_conv_fwd_stats', xd, wd, None, rd, y1, N, H, H, C, C, O, O, O if res else 0, K, K, 1, K // 2, 1, 0, part, hip.stream())
    assert torch.equal(y0, y1)
    yf = y1.double().reshape(M, O)
    s, q = part[:, 0].double().sum(0), part[:, 1].double().sum(0)
    assert_close(s.cpu().numpy(), yf.sum(0).cpu().numpy(), 1e-5, 1e-3, 'sum')
    assert_close(q.cpu().numpy(), (yf * yf).sum(0).cpu().numpy(), 1e-5, 1e-3, 'sum of squares')
    # per tile too (rows of tile t)
    t = nblk // 2
    rows = yf[t * int(bm):(t + 1) * int(bm)]
    assert_close(part[t, 0].cpu().numpy(), rows.sum(0).cpu().numpy(), 1e-5, 1e-3, 'tile sum')
    # finalize over these partials == finalize over sn_bn_stats partials
    ws = torch.empty(hip.query('sn_bn_workspace_bytes', M, O), dtype=torch.uint8, device=dev())
    f = lambda v: torch.full((O,), v, device=dev())
    outs = []
    for use_blocks in (False, True):
        g, b, rm, rv, sc, sh, sm, si = f(1.3), f(0.2), f(0.0), f(1.0), f(0.0), f(0.0), f(0.0), f(0.0)
        if use_blocks:
            hip.call('sn_bn_finalize_blocks', part, nblk, M, O, 2e-5, 0.9, g, b, rm, rv, sc, sh, sm, si, hip.stream())
        else:
            hip.call('sn_bn_stats', y1, M, O, O, ws, hip.stream())
            hip.call('sn_bn_finalize', ws, M, O, 2e-5, 0.9, g, b, rm, rv, sc, sh, sm, si, hip.stream())
        outs.append([t_.clone() for t_ in (sc, sh, sm, si, rm, rv)])
    for a, b in zip(*outs):
        assert_close(a.cpu().numpy(), b.cpu().numpy(), 1e-5, 1e-6, 'finalize')
    # a narrow layer does not qualify
    assert hip.query('sn_conv_fwd_stats_blocks', N, H, H, C, C, 48, 48, 0, 1, 1, 1, 0, 1) == 0


@pytest.mark.parametrize('N,H,C,O,K,bm,act', [(3, 17, 128, 192, 1, 'dma6', 1), (2, 24, 128, 256, 3, 'dma16', 1), (4, 12, 192, 64, 3, 'dma5', 0),
                                              (2, 16, 320, 128, 1, 'dma14', 2), (2, 24, 128, 256, 3, 'dma14', 1), (3, 17, 128, 192, 1, 'dma5', 1),
                                              (2, 24, 128, 256, 3, 'dma18', 1), (3, 17, 128, 192, 1, 'dma18', 2)])
def test_conv_dgrad_bn_epilogue(N, H, C, O, K, bm, act, monkeypatch, conv_tuning):
    """sn_conv_dgrad_bn: same dx as sn_conv_dgrad, and partials that make sn_bn_backward_blocks reproduce sn_bn_backward
    (dx of the BatchNorm below, dgamma, dbeta) -- ReLU / none / ReLU6 masks, several tile configurations, ragged tiles."""
    hip = _hip()
    hip.call('sn_conv_tune', int(bm[3:]))
    bm = str(CONV_DMA_BM[int(bm[3:])])
    rs = np.random.RandomState(N + H + C)
    M = N * H * H
    dy = torch.from_numpy(rs.standard_normal((N, H, H, O)).astype(np.float32)).to(dev()).half()
    wt = torch.from_numpy((rs.standard_normal((C, K * K, O)) / np.sqrt(K * K * O)).astype(np.float32)).to(dev()).half()
    bnx = torch.from_numpy(rs.standard_normal((N, H, H, C)).astype(np.float32)).to(dev()).half()
    f = lambda a: torch.from_numpy(a.astype(np.float32)).to(dev())
    scale, shift = f(rs.uniform(0.5, 1.5, C)), f(rs.uniform(-0.5, 2.5, C))
    mean, invstd = f(rs.standard_normal(C) * 0.1), f(rs.uniform(0.5, 2, C))
    dx0 = torch.empty((N, H, H, C), dtype=torch.float16, device=dev())
    hip.call('sn_conv_dgrad', dy, wt, None, dx0, N, H, H, C, C, O, O, 0, K, K, 1, K // 2, 1, 0, hip.stream())
    nblk = hip.query('sn_conv_dgrad_bn_blocks', N, H, H, C, C, O, O, 0, K, K, 1, K // 2, 1)
    assert nblk == -(-M // int(bm))
    part = torch.full((nblk, 2, C), 7.0, dtype=torch.float32, device=dev())
    dx1 = torch.empty_like(dx0)
    hip.call('sn_conv_dgrad_bn', dy, wt, None, dx1, N, H, H, C, C, O, O, 0, K, K, 1, K // 2, 1, bnx, C, scale, shift, mean, act, part,
             hip.stream())
    assert torch.equal(dx0, dx1)
    ws = torch.empty(hip.query('sn_bn_workspace_bytes', M, C), dtype=torch.uint8, device=dev())
    res = []
    for blocks in (False, True):
        dg, db = torch.zeros(C, device=dev()), torch.zeros(C, device=dev())
        out = torch.empty_like(dx0)
        if blocks:
            hip.call('sn_bn_backward_blocks', part, nblk, dx1, bnx, None, out, M, C, C, C, C, C, scale, shift, mean, invstd, act, ws, dg, db,
                     hip.stream())
        else:
            hip.call('sn_bn_backward', dx1, bnx, None, out, M, C, C, C, C, C, scale, shift, mean, invstd, act, ws, dg, db, hip.stream())
        res.append((out.float().cpu().numpy(), dg.cpu().numpy(), db.cpu().numpy()))
    (o0, g0, b0), (o1, g1, b1) = res
    assert_close(g1, g0, 1e-4, 1e-4 * np.abs(g0).max(), 'dgamma')
    assert_close(b1, b0, 1e-4, 1e-4 * np.abs(b0).max(), 'dbeta')
    assert_close(o1, o0, 2e-3, 2e-3 * np.abs(o0).max(), 'bn dx')
    assert np.abs(g0).max() > 0


@pytest.mark.parametrize('N,C,H,W,O,K,pad', [(2, 256, 32, 32, 128, 1, 0), (3, 128, 40, 72, 64, 3, 1), (20, 256, 64, 64, 256, 3, 1),
                                             (20, 512, 64, 64, 1024, 1, 0), (2, 128, 24, 16, 192, 3, 1), (1, 64, 6, 10, 128, 5, 2)])
def test_stride2_dgrad_by_parity_class_equals_every_tap_walk(N, C, H, W, O, K, pad, conv_tuning):
    """The data gradient of a stride-2 convolution enumerated parity class by parity class (each class only the taps that reach it:
    ConvParams::cls) against the walk over every tap for every pixel (sn_conv_dgrad_by_class(0)): dx BIT-equal (the skipped taps
    contributed exact zeros), with an accumulate operand, through the fused BatchNorm-backward epilogue (per-channel sums of the
    partials equal; the row tiles are grouped differently), for the 1x1 shortcut (three classes receive no tap at all), 3x3 and 5x5
    kernels, the C2 launch shapes of stage3_unit1, and every tile configuration that takes such layers."""
    hip = _hip()
    rs = np.random.RandomState(N * C + K)
    Ho, Wo = (H + 2 * pad - K) // 2 + 1, (W + 2 * pad - K) // 2 + 1
    Op = (O + 7) // 8 * 8
    dy = torch.zeros((N, Ho, Wo, Op), dtype=torch.float16, device=dev())
    dy[..., :O] = torch.from_numpy(rs.standard_normal((N, Ho, Wo, O)).astype(np.float32)).to(dev()).half()
    wt = torch.zeros((C, K * K, Op), dtype=torch.float16, device=dev())
    wt[..., :O] = torch.from_numpy((rs.standard_normal((C, K * K, O)) / np.sqrt(K * K * O)).astype(np.float32)).to(dev()).half()
    acc = torch.from_numpy(rs.standard_normal((N, H, W, C)).astype(np.float32)).to(dev()).half()
    bnx = torch.from_numpy(rs.standard_normal((N, H, W, C)).astype(np.float32)).to(dev()).half()
    f = lambda a: torch.from_numpy(a.astype(np.float32)).to(dev())
    scale, shift, mean = f(rs.uniform(0.5, 1.5, C)), f(rs.uniform(-0.5, 0.5, C)), f(rs.standard_normal(C) * 0.1)
    geom = (N, H, W, C, C, Op, Op)
    try:
        for cfg in ((-1, 14, 16, 5) if N <= 3 else (-1,)):
            hip.call('sn_conv_tune', cfg)
            res = {}
            for on in (0, 1):
                hip.call('sn_conv_dgrad_by_class', on)
                dx = torch.full((N, H, W, C), 7.0, dtype=torch.float16, device=dev())
                hip.call('sn_conv_dgrad', dy, wt, None, dx, *geom, 0, K, K, 2, pad, 1, 0, hip.stream())
                dxa = torch.full((N, H, W, C), 7.0, dtype=torch.float16, device=dev())
                hip.call('sn_conv_dgrad', dy, wt, acc, dxa, *geom, C, K, K, 2, pad, 1, 0, hip.stream())
                nblk = hip.query('sn_conv_dgrad_bn_blocks', *geom, 0, K, K, 2, pad, 1)
                part, dxb = None, None
                if nblk > 0:
                    part = torch.full((nblk, 2, C), 7.0, dtype=torch.float32, device=dev())
                    dxb = torch.full((N, H, W, C), 7.0, dtype=torch.float16, device=dev())
                    hip.call('sn_conv_dgrad_bn', dy, wt, None, dxb, *geom, 0, K, K, 2, pad, 1, bnx, C, scale, shift, mean, 1, part,
                             hip.stream())
                torch.cuda.synchronize()
                res[on] = (dx, dxa, dxb, part, nblk)
            (dx0, dxa0, dxb0, p0, n0), (dx1, dxa1, dxb1, p1, n1) = res[0], res[1]
            assert torch.equal(dx0, dx1) and torch.equal(dxa0, dxa1), (cfg, 'dx')
            assert float(dx1.float().abs().max()) > 0
            if cfg == -1 and H % 2 == 0 and W % 2 == 0:
                assert (n0 > 0) == (n1 > 0)
            if n0 > 0 and n1 > 0:
                assert torch.equal(dxb0, dxb1) and torch.equal(dxb1, dx1), (cfg, 'dx of the fused entry')
                assert n1 >= n0                                  # four classes, each with its own ragged last tile
                s0, s1 = p0.double().sum(0).cpu().numpy(), p1.double().sum(0).cpu().numpy()
                assert_close(s1, s0, 1e-5, 1e-4 * np.abs(s0).max(), 'BatchNorm-backward sums by class')
    finally:
        hip.call('sn_conv_dgrad_by_class', 1)
    # the oracle on the smallest shapes (the every-tap walk has its own oracle tests: test_conv_dgrad_wgrad)
    if N <= 3:
        xt = torch.zeros((N, C, H, W), requires_grad=True)
        w_oihw = wt[..., :O].float().cpu().permute(2, 0, 1).reshape(O, C, K, K)
        Fnn.conv2d(xt, w_oihw, None, 2, pad).backward(dy[..., :O].float().cpu().permute(0, 3, 1, 2))
        want = xt.grad.numpy()
        assert_close(from_nhwc(dx1), want, 1e-2, 1e-2 * np.abs(want).max(), 'dgrad by class vs torch')


@pytest.mark.parametrize('rows,cols,in_ld,out_ld', [(1, 15728640, 15728640, 15728640), (5000, 1024, 1024, 3072), (333, 72, 80, 72), (7, 30, 33, 40),
                                                    (1, 1000, 1000, 1000), (64, 8, 24, 16)])
def test_copy2d_vectorised_and_scalar_paths(rows, cols, in_ld, out_ld):
    """sn_copy2d (Concat slices, Cast, the step's input copies): the 8-elements-per-thread path (row length and pitches multiples
    of 8 on 16-byte aligned pointers; a single row is a flat copy) and the scalar path, all four dtype pairs, against torch."""
    hip = _hip()
    rs = np.random.RandomState(rows + cols)
    for in_dt, out_dt in ((0, 0), (0, 1), (1, 0), (1, 1)):
        ti, to = (torch.float16, torch.float32)[in_dt], (torch.float16, torch.float32)[out_dt]
        src = torch.from_numpy(rs.standard_normal((rows, in_ld)).astype(np.float32)).to(dev()).to(ti)
        dst = torch.full((rows, out_ld), 7.0, dtype=to, device=dev())
        hip.call('sn_copy2d', src, dst, rows, cols, in_ld, out_ld, in_dt, out_dt, hip.stream())
        want = torch.full((rows, out_ld), 7.0, dtype=to, device=dev())
        want[:, :cols] = src[:, :cols].to(to)
        assert torch.equal(dst, want), (rows, cols, in_dt, out_dt)


@pytest.mark.parametrize('N,C,H,W,O,K,pad,dil,hb,hr,relu,split', [
    (2, 256, 36, 36, 256, 3, 1, 1, True, False, 1, True), (2, 1024, 36, 36, 256, 1, 0, 1, True, False, 1, True),
    (2, 256, 36, 36, 1024, 1, 0, 1, False, True, 0, False),           # four K-steps: too short a contraction to split
    (2, 512, 36, 40, 256, 3, 2, 2, True, True, 0, True),              # dilated, residual in the reduce
    (1, 3072, 30, 44, 512, 3, 1, 1, True, False, 1, True),            # the RPN convolution of one test image
    (2, 3072, 36, 36, 512, 3, 1, 1, True, False, 1, True),            # ... of a 2-chip batch: 164 tiles, 432 K-steps
    (2592, 4608, 1, 1, 512, 1, 0, 1, True, False, 0, True),           # the deformable GEMM of that batch: 164 tiles, 72 K-steps
    (600, 12544, 1, 1, 1024, 1, 0, 1, True, False, 1, True),          # fc_new_1 over 600 RoIs: planned 128 x 256, split as 64 x 128
    (600, 12544, 1, 1, 98, 1, 0, 1, True, False, 0, True),            # the offset FullyConnected: 98 channels (element-wise reduce), 10 tiles
    (20, 256, 32, 32, 256, 3, 1, 1, True, False, 1, False)])          # a training-size launch: enough tiles
def test_conv_fwd_splitk_equals_plain_forward(N, C, H, W, O, K, pad, dil, hb, hr, relu, split):
    """sn_conv_fwd_splitk (test-time launches with far fewer output tiles than CUs: contraction split over copies of the tile grid,
    fp32 partials reduced in order with the bias / residual / ReLU epilogue) against sn_conv_fwd and against torch-CPU fp32, at the
    shapes of a 2-chip batch of the finest test scale; a launch with enough tiles (the last case) reports 0 bytes and runs the
    plain kernel."""
    hip = _hip()
    rs = np.random.RandomState(N * C + O)
    x = rs.standard_normal((N, C, H, W)).astype(np.float32)
    w = (rs.standard_normal((O, C, K, K)) / np.sqrt(C * K * K)).astype(np.float32)
    b = rs.standard_normal(O).astype(np.float32) if hb else None
    res = rs.standard_normal((N, O, H, W)).astype(np.float32) if hr else None
    xd = to_nhwc_f16(x)
    wd = torch.from_numpy(w_to_otI(w)).to(dev()).half().contiguous()
    bd = torch.from_numpy(b).to(dev()) if hb else None
    rd = to_nhwc_f16(res) if hr else None
    geom = (N, H, W, C, C, O, O, O if hr else 0, K, K, 1, pad, dil)
    need = hip.query('sn_conv_fwd_splitk_workspace_bytes', *geom)
    assert (need > 0) == split, need
    y0 = torch.full((N, H, W, O), 7.0, dtype=torch.float16, device=dev())
    hip.call('sn_conv_fwd', xd, wd, bd, rd, y0, *geom, relu, 0, hip.stream())
    ws = torch.full((max(need, 16),), 0x7f, dtype=torch.uint8, device=dev())
    y1 = torch.full((N, H, W, O), 7.0, dtype=torch.float16, device=dev())
    hip.call('sn_conv_fwd_splitk', xd, wd, bd, rd, y1, *geom, relu, ws, need, hip.stream())
    torch.cuda.synchronize()
    want = _ref_conv(x, w, b, res, 1, pad, dil, relu)
    assert_close(from_nhwc(y1), want, 1e-2, 1e-2 * np.abs(want).max(), 'split-K forward vs torch')
    # same operands, fp32 accumulation on both sides, one fp16 rounding at the end: the two kernels differ by summation order only
    assert_close(y1.float().cpu().numpy(), y0.float().cpu().numpy(), 2e-3, 2e-3 * float(y0.float().abs().max()), 'split-K vs plain')
    if need == 0:
        assert torch.equal(y0, y1)
    y2 = torch.full((N, H, W, O), 7.0, dtype=torch.float16, device=dev())
    hip.call('sn_conv_fwd_splitk', xd, wd, bd, rd, y2, *geom, relu, ws, need, hip.stream())
    assert torch.equal(y1, y2)          # deterministic
    if not hr:                          # the fp32-output entry (no residual): against sn_conv_fwd's out_f32 and the fp16 result
        f0 = torch.full((N, H, W, O), 7.0, dtype=torch.float32, device=dev())
        f1 = torch.full((N, H, W, O), 7.0, dtype=torch.float32, device=dev())
        hip.call('sn_conv_fwd', xd, wd, bd, None, f0, *geom, relu, 1, hip.stream())
        hip.call('sn_conv_fwd_splitk_f32', xd, wd, bd, f1, N, H, W, C, C, O, O, K, K, 1, pad, dil, relu, ws, need, hip.stream())
        torch.cuda.synchronize()
        assert_close(f1.cpu().numpy(), f0.cpu().numpy(), 1e-4, 1e-4 * float(f0.abs().max()), 'split-K fp32 vs plain fp32')
        assert torch.equal(f1.half(), y1)
        if need == 0:
            assert torch.equal(f0, f1)


def test_weight_transpose_batched_equals_single():
    """sn_weight_transpose_batched (one launch, LDS-tiled) against sn_weight_transpose per weight: ragged O / I, taps, O_pad."""
    hip = _hip()
    rs = np.random.RandomState(31)
    shapes = [(256, 9, 256), (42, 1, 512), (1024, 1, 64), (81, 49, 24), (7, 3, 200), (512, 9, 3072 // 8)]
    srcs = [torch.from_numpy(rs.standard_normal(sh).astype(np.float32)).to(dev()) for sh in shapes]
    pad8 = lambda n: (n + 7) // 8 * 8
    want, dsts = [], []
    rec = np.zeros(len(shapes), dtype=np.dtype([('src', '<u8'), ('dst', '<u8'), ('O', '<i4'), ('T', '<i4'), ('I', '<i4'), ('Opad', '<i4'),
                                                ('tile0', '<i4'), ('tiles_o', '<i4'), ('tiles_i', '<i4'), ('pad', '<i4')]))
    tile0 = 0
    for k, ((o, t, i), src) in enumerate(zip(shapes, srcs)):
        w = torch.full((i, t, pad8(o)), 7.0, dtype=torch.float16, device=dev())
        hip.call('sn_weight_transpose', src, w, o, t, i, pad8(o), hip.stream())
        want.append(w)
        d = torch.full((i, t, pad8(o)), 7.0, dtype=torch.float16, device=dev())
        dsts.append(d)
        to, ti = (pad8(o) + 63) // 64, (i + 63) // 64
        rec[k] = (src.data_ptr(), d.data_ptr(), o, t, i, pad8(o), tile0, to, ti, 0)
        tile0 += t * to * ti
    desc = torch.from_numpy(rec.view(np.uint8).copy()).to(dev())
    hip.call('sn_weight_transpose_batched', desc, len(shapes), tile0, hip.stream())
    torch.cuda.synchronize()
    for (o, t, i), w, d, src in zip(shapes, want, dsts, srcs):
        assert torch.equal(w, d), (o, t, i)
        assert torch.equal(d[:, :, :o].float(), src.half().float().permute(2, 1, 0))
        assert float(d[:, :, o:].abs().sum()) == 0.0


@pytest.mark.parametrize('Fh,Fw', [(16, 16), (12, 20)])
def test_multi_proposal_target_vs_oracle(Fh, Fw):
    """Square training chips and the non-square feature maps of test images."""
    hip = _hip()
    rs = np.random.RandomState(8)
    B, A, stride, pre, post, G = 2, 21, 16, 600, 50, 100
    cls_prob = rs.uniform(0, 1, (B, 2, A * Fh, Fw)).astype(np.float32)
    bbox_pred = (rs.standard_normal((B, 4 * A, Fh, Fw)) * 0.3).astype(np.float32)
    im_info = np.array([[Fh * 16, Fw * 16, 1.0], [Fh * 16, Fw * 16, 1.6]], np.float32)
    gt = -np.ones((B, G, 5), np.float32)
    for b in range(B):
        n = 12
        c = rs.uniform(30, min(Fh, Fw) * 16 - 30, (n, 2))
        wh = rs.uniform(20, 150, (n, 2))
        gt[b, :n, :4] = np.concatenate((c - wh / 2, c + wh / 2), 1)
        gt[b, :n, 4] = rs.randint(1, 81, n)
    vr = np.array([[0, 256], [40, 120]], np.float32)
    from sniper_amd.data.anchors import generate_anchors
    scales, ratios = (2, 4, 7, 10, 13, 16, 24), (0.5, 1, 2)
    base = generate_anchors(stride, list(ratios), np.array(scales, np.float32)).astype(np.float32)
    td = lambda z: torch.from_numpy(np.ascontiguousarray(z)).to(dev())
    ws = torch.empty(hip.query('sn_proposal_workspace_bytes', B, A, Fh, Fw, pre, post), dtype=torch.uint8, device=dev())
    rois = torch.empty((B * post, 5), device=dev())
    label = torch.empty((B * post,), device=dev())
    tgt, wgt = torch.empty((B * post, 4), device=dev()), torch.empty((B * post, 4), device=dev())
    stds = np.array([0.1, 0.1, 0.2, 0.2], np.float32)
    hip.call('sn_multi_proposal_target', td(cls_prob), td(bbox_pred), td(im_info), td(gt), td(vr), td(base), B, A, Fh, Fw, stride, G,
             pre, post, 0.7, 0.0, 0.5, stds.ctypes.data, ws, rois, label, tgt, wgt, hip.stream())
    torch.cuda.synchronize()
    want_rois, want_scores, dbg = onn.proposals(cls_prob, bbox_pred, im_info, stride, scales, ratios, pre, post, 0.7, 0.0)
    got_rois = rois.cpu().numpy()
    # the index sets (sort order, NMS survivors) agree; a decoded coordinate may differ in its last float32 bit
    n_exact = (got_rois == want_rois).all(1).mean()
    assert n_exact > 0.98, n_exact
    assert_close(got_rois, want_rois, 1e-5, 1e-3, 'rois')
    wl, wt, ww = onn.proposal_targets(got_rois, gt, vr, post)
    assert np.array_equal(label.cpu().numpy(), wl)
    assert np.array_equal(wgt.cpu().numpy(), ww)
    assert_close(tgt.cpu().numpy(), wt, 1e-5, 1e-5, 'bbox targets')
    assert (wl > 0).sum() > 0 and (wl == -1).sum() >= 0
    # test-time op: same RoIs + their scores, cls_prob viewed as (B, 2A, Fh, Fw)
    rois2, sc = torch.empty_like(rois), torch.empty((B * post,), device=dev())
    hip.call('sn_multi_proposal', td(cls_prob), td(bbox_pred), td(im_info), td(base), B, A, Fh, Fw, stride, pre, post, 0.7, 0.0, ws,
             rois2, sc, hip.stream())
    assert torch.equal(rois2, rois)
    assert_close(sc.cpu().numpy(), want_scores, 1e-6, 1e-6, 'roi scores')


def _ulp_close(got, want, ulps=1):
    """every element within `ulps` float32 units in the last place"""
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    return np.abs(got.astype(np.float64) - want.astype(np.float64)) <= ulps * np.spacing(np.maximum(np.abs(got), np.abs(want)))


def c2_proposal_inputs(seed=5):
    """MultiProposalTarget / MultiProposal inputs at the BASELINE C2 launch shape (symbols/faster/resnet_mx_101_e2e.py:283-284,
    347-355: B = 20 chips, A = 21 anchors, 32 x 32 map, 6000 -> 300, 100 x 5 ground truth), every chip a different regime:
      0-7   continuous scores, moderate deltas (the usual case: 300 survivors, some decoded coordinates clipped)
      8-11  scores quantised to 1/64 (hundreds of exact ties at the 6000 boundary and inside it: order = anchor index)
      12    every box decodes to the whole chip (one NMS survivor, repeated cyclically 300 times)
      13-14 large boxes around few centres (fewer than 300 survivors: cyclic padding with a remainder)
      15-17 small boxes + rpn_min_size 16 at scale 1 / 1.667 / 2.917 (rejections get score -1 and sort last)
      18    no ground truth at all, 19: 100 ground-truth boxes (full table), duplicated boxes (arg-max = first)
    valid ranges are the three SNIPER scales' (MNIteratorE2E.py:158-162 applied to yml:98-101), so fg / ignore / bg all occur."""
    rs = np.random.RandomState(seed)
    B, A, F, G = 20, 21, 32, 100
    p1 = rs.uniform(0, 1, (B, 1, A * F, F))
    p1[8:12] = np.round(p1[8:12] * 64) / 64
    cls_prob = np.concatenate((1 - p1, p1), 1).astype(np.float32)
    bbox_pred = (rs.standard_normal((B, 4 * A, F, F)) * 0.4).astype(np.float32)
    d = bbox_pred.reshape(B, A, 4, F, F)
    d[12, :, 2:] = 6.0
    d[13:15, :, 2:] = 2.6 + 0.3 * rs.standard_normal((2, A, 2, F, F)).astype(np.float32)
    d[13:15, :, :2] *= 0.05
    d[15:18, :, 2:] -= 1.2
    scales = np.array([1.0] * 15 + [1.0, 1.667, 2.917] + [1.0, 1.0], np.float32)
    im_info = np.stack((np.full(B, 512.0), np.full(B, 512.0), scales), 1).astype(np.float32)
    gt = -np.ones((B, G, 5), np.float32)
    for b in range(B):
        n = 0 if b == 18 else (100 if b == 19 else int(rs.randint(1, 60)))
        c = rs.uniform(20, 490, (n, 2))
        wh = np.exp(rs.uniform(np.log(8), np.log(400), (n, 2)))
        gt[b, :n, :4] = np.round(np.clip(np.concatenate((c - wh / 2, c + wh / 2), 1), 0, 511))
        gt[b, :n, 4] = rs.randint(1, 81, n)
    gt[19, 50:60, :4] = gt[19, 40:50, :4]          # equal IoU with two rows: the first one is the match
    ranges = [(0.0, 80.0), (32.0 * 1.667, 150.0 * 1.667), (120.0 * 2.917, 512.0)]
    vr = np.array([ranges[b % 3] for b in range(B)], np.float32)
    vr[15], vr[16], vr[17] = ranges[0], ranges[1], ranges[2]
    return cls_prob, bbox_pred, im_info, gt, vr


@pytest.mark.parametrize('min_size', [0.0, 16.0])
def test_multi_proposal_target_at_the_c2_launch_shape_vs_oracle(min_size):
    """sn_multi_proposal_target and sn_multi_proposal against oracle/nn.py at the BASELINE C2 launch shape (B = 20, 21 504
    anchors per chip, 6000 -> 300, G = 100): the RoI INDEX SETS must be those of the oracle -- same survivors in the same order,
    i.e. every output row is the oracle's row (scores are copied, so bit-equal; a decoded coordinate may differ in its last
    float32 bit: device exp vs libm) -- and labels / weights bit-equal, targets to 1e-5 (log of the device)."""
    hip = _hip()
    cls_prob, bbox_pred, im_info, gt, vr = c2_proposal_inputs()
    B, A, F, G, stride, pre, post = 20, 21, 32, 100, 16, 6000, 300
    from sniper_amd.data.anchors import generate_anchors
    scales, ratios = (2, 4, 7, 10, 13, 16, 24), (0.5, 1, 2)
    base = generate_anchors(stride, list(ratios), np.array(scales, np.float32)).astype(np.float32)
    td = lambda z: torch.from_numpy(np.ascontiguousarray(z)).to(dev())
    ws = torch.empty(hip.query('sn_proposal_workspace_bytes', B, A, F, F, pre, post), dtype=torch.uint8, device=dev())
    rois = torch.full((B * post, 5), 7.0, device=dev())
    label = torch.full((B * post,), 7.0, device=dev())
    tgt, wgt = torch.full((B * post, 4), 7.0, device=dev()), torch.full((B * post, 4), 7.0, device=dev())
    stds = np.array([0.1, 0.1, 0.2, 0.2], np.float32)
    hip.call('sn_multi_proposal_target', td(cls_prob), td(bbox_pred), td(im_info), td(gt), td(vr), td(base), B, A, F, F, stride, G,
             pre, post, 0.7, min_size, 0.5, stds.ctypes.data, ws, rois, label, tgt, wgt, hip.stream())
    rois2, sc = torch.full((B * post, 5), 7.0, device=dev()), torch.full((B * post,), 7.0, device=dev())
    hip.call('sn_multi_proposal', td(cls_prob), td(bbox_pred), td(im_info), td(base), B, A, F, F, stride, pre, post, 0.7, min_size, ws,
             rois2, sc, hip.stream())
    torch.cuda.synchronize()
    want_rois, want_scores, dbg = onn.proposals(cls_prob, bbox_pred, im_info, stride, scales, ratios, pre, post, 0.7, min_size)
    got_rois, got_sc = rois.cpu().numpy(), sc.cpu().numpy()
    assert torch.equal(rois2, rois)
    # scores are the input probabilities (or the -1 of a min_size rejection) of the selected anchors: bit-equal, row for row
    assert np.array_equal(got_sc, want_scores)
    assert np.array_equal(got_rois[:, 0], want_rois[:, 0])
    ok = _ulp_close(got_rois[:, 1:], want_rois[:, 1:], 1)
    assert ok.all(), 'RoI rows differ from the oracle beyond 1 ulp: rows %s' % np.where(~ok.all(1))[0][:10]
    assert (got_rois == want_rois).all(1).mean() > 0.98
    # the regimes the inputs were built for did occur
    nkeep = np.array([len(k) for _, _, k in dbg])
    assert nkeep[12] == 1 and (nkeep[13:15] < post).all() and (nkeep[13:15] > 1).all() and (nkeep[:8] >= post).all(), nkeep
    for b in (12, 13, 14):          # cyclic repetition of the survivors
        r = got_rois[b * post:(b + 1) * post]
        assert np.array_equal(r, r[np.arange(post) % nkeep[b]])
    for b in range(8, 12):          # ties at and inside the cut: the 6000 selected are the stable-sort prefix
        sb = dbg[b][1]
        assert (np.diff(sb[:, 4]) == 0).sum() > 1000
    if min_size > 0:
        rej = [int((dbg[b][1][:, 4] == -1).sum()) for b in (15, 16, 17)]
        assert min(rej) >= 0 and max(rej) > 0, rej
    # labels / targets: against the oracle on the DEVICE's rois (identical inputs) and on the oracle's own rois (whole op)
    wl, wt, ww = onn.proposal_targets(got_rois, gt, vr, post)
    gl = label.cpu().numpy()
    assert np.array_equal(gl, wl) and np.array_equal(wgt.cpu().numpy(), ww)
    assert_close(tgt.cpu().numpy(), wt, 1e-5, 1e-5, 'bbox targets')
    wl2, wt2, ww2 = onn.proposal_targets(want_rois, gt, vr, post)
    assert (gl != wl2).mean() < 1e-3          # a 1-ulp coordinate can move an IoU across 0.5 for a handful of RoIs at most
    assert (gl > 0).sum() > 30 and (gl == -1).sum() > 20 and (gl == 0).sum() > 1000
    assert (gl[18 * post:19 * post] == 0).all()          # no ground truth: background only
    assert (gl[19 * post:] > 0).any()


@pytest.mark.parametrize('A,Fh,Fw,pre,quant', [(21, 32, 32, 6000, 0), (21, 32, 32, 6000, 64), (15, 16, 16, 6000, 8), (21, 40, 56, 6000, 0),
                                                (3, 4, 5, 20, 4), (21, 50, 80, 6000, 16), (21, 88, 125, 6000, 0)])
def test_proposal_topk_select_equals_full_sort(A, Fh, Fw, pre, quant, monkeypatch):
    """The proposal ordering (radix select of the pre_nms_top_n best keys + LDS sort) must produce exactly the RoIs of the
    full sort: R101 training size, heavily tied scores (quantised probabilities, min_size rejections -> score -1),
    pre >= total (MobileNetV2: every anchor selected), a non-square test-image map, a tiny map, and the 84 k / 231 k-anchor maps of the
    800 x 1280 and 1400 x 2000 test scales (streaming form of the select: eight loads in flight per thread, ragged last group)."""
    hip = _hip()
    rs = np.random.RandomState(A * Fh + quant)
    B, stride, post = 3, 16, 300
    p1 = rs.uniform(0, 1, (B, 1, A * Fh, Fw))
    if quant:
        p1 = np.round(p1 * quant) / quant
    cls_prob = np.concatenate((1 - p1, p1), 1).astype(np.float32)
    bbox_pred = (rs.standard_normal((B, 4 * A, Fh, Fw)) * 0.5).astype(np.float32)
    im_info = np.array([[Fh * 16, Fw * 16, 1.0]] * B, np.float32)
    from sniper_amd.data.anchors import generate_anchors
    scales = (2, 4, 7, 10, 13, 16, 24)[:A // 3]
    base = generate_anchors(stride, [0.5, 1, 2], np.array(scales, np.float32)).astype(np.float32)
    td = lambda z: torch.from_numpy(np.ascontiguousarray(z)).to(dev())
    post = min(post, pre, A * Fh * Fw)
    ws = torch.empty(hip.query('sn_proposal_workspace_bytes', B, A, Fh, Fw, pre, post), dtype=torch.uint8, device=dev())
    outs = []
    for full in ('1', None):
        hip.call('sn_debug_option', b'proposal_full_sort', 1 if full else 0)
        rois, sc = torch.empty((B * post, 5), device=dev()), torch.empty((B * post,), device=dev())
        hip.call('sn_multi_proposal', td(cls_prob), td(bbox_pred), td(im_info), td(base), B, A, Fh, Fw, stride, pre, post, 0.7, 24.0, ws,
                 rois, sc, hip.stream())
        torch.cuda.synchronize()
        outs.append((rois.clone(), sc.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float(outs[1][1].max()) > 0


@pytest.mark.parametrize('B,C,H,W,R,SC', [(2, 64, 12, 12, 9, 16), (3, 256, 10, 14, 40, 16), (2, 128, 16, 16, 600, 32)])
def test_dpsroi_pool_fwd_bwd_vs_oracle(B, C, H, W, R, SC):
    hip = _hip()
    rs = np.random.RandomState(9)
    P, S = 7, 4
    data = rs.standard_normal((B, C, H, W)).astype(np.float32)
    rois = np.zeros((R, 5), np.float32)
    rois[:, 0] = rs.randint(0, B, R)
    c = rs.uniform(20, SC * min(H, W) - 20, (R, 2))
    wh = rs.uniform(4, 120 * SC / 16, (R, 2))
    rois[:, 1:3], rois[:, 3:5] = c - wh / 2, c + wh / 2
    rois[0, 1:] = [-30, -20, 40, 50]   # partly outside
    rois[1, 1:] = [0, 0, SC * W - 1, SC * H - 1]   # whole map
    rois[2, 1:] = [33, 47, 34, 48]   # tiny: every sample of a bin in one cell
    trans = (rs.standard_normal((R, 2, P, P)) * 0.5).astype(np.float32)
    dd = to_nhwc_f16(data)
    td = lambda z: torch.from_numpy(z).to(dev())
    ws = torch.empty(hip.query('sn_dpsroi_bwd_workspace_bytes', R), dtype=torch.uint8, device=dev())
    for tr, tstd in ((None, 0.0), (trans, 0.1)):
        out = torch.empty((R, P, P, C), dtype=torch.float16, device=dev())
        hip.call('sn_dpsroi_pool_fwd', dd, td(rois), None if tr is None else td(tr), out, R, H, W, C, P, S, 1.0 / SC, tstd, hip.stream())
        want = onn.dpsroi_pool(f16r(data).astype(np.float64), rois, tr, P, S, 1.0 / SC, tstd)
        assert_close(out.float().cpu().numpy().transpose(0, 3, 1, 2), want, 1e-2, 1e-2, 'dpsroi fwd')
        # the (image, 64-channel slab)-stationary kernel (taken with B known, C % 64 == 0, R >= 8 B): the per-RoI kernel's outputs
        out_s = torch.full((R, P, P, C), 7.0, dtype=torch.float16, device=dev())
        hip.call('sn_dpsroi_pool_fwd_images', dd, td(rois), None if tr is None else td(tr), out_s, R, B, H, W, C, P, S, 1.0 / SC, tstd,
                 hip.stream())
        assert torch.equal(out_s, out)
        dout = rs.standard_normal((R, C, P, P)).astype(np.float32)
        dod = torch.from_numpy(np.ascontiguousarray(dout.transpose(0, 2, 3, 1))).to(dev()).half()
        wd, wtr = onn.dpsroi_pool_backward(f16r(dout).astype(np.float64), f16r(data).astype(np.float64), rois, tr, P, S, 1.0 / SC, tstd)
        for f32 in (1, 0):
            # poisoned outputs: the kernels must overwrite every element (no zeroing contract)
            d_data = torch.full((B, H, W, C), 7.0, dtype=torch.float32 if f32 else torch.float16, device=dev())
            d_trans = torch.full((R, 2, P, P), 7.0, dtype=torch.float32, device=dev())
            hip.call('sn_dpsroi_pool_bwd', dod, dd, td(rois), None if tr is None else td(tr), d_data, f32,
                     d_trans if tr is not None else None, R, B, H, W, C, P, S, 1.0 / SC, tstd, ws, hip.stream())
            tol = 1e-3 if f32 else 1e-2
            assert_close(d_data.float().cpu().numpy().transpose(0, 3, 1, 2), wd, tol, tol * np.abs(wd).max(), 'dpsroi d_data')
            if tr is not None:
                assert_close(d_trans.cpu().numpy(), wtr, 1e-3, 1e-3 * np.abs(wtr).max(), 'dpsroi d_trans')
        # the tile kernel sums in a fixed order: bit-reproducible
        d2 = torch.empty_like(d_data)
        hip.call('sn_dpsroi_pool_bwd', dod, dd, td(rois), None if tr is None else td(tr), d2, 0,
                 d_trans if tr is not None else None, R, B, H, W, C, P, S, 1.0 / SC, tstd, ws, hip.stream())
        assert torch.equal(d2, d_data)


@pytest.mark.parametrize('B,D,G,P,H,W,R,SC', [(2, 5, 3, 3, 12, 12, 9, 16), (2, 81, 7, 7, 10, 14, 40, 16), (2, 8, 7, 7, 16, 16, 150, 16),
                                              (1, 300, 2, 4, 9, 8, 20, 16)])
def test_position_sensitive_pool_fwd_bwd_vs_oracle(B, D, G, P, H, W, R, SC):
    """sn_psroi_pool_* (group_size G > 1, the R-FCN head of BASELINE config C4) against oracle/nn.py; D = 81 and 8 are
    the class / box map depths, (G, P) = (2, 4) covers several bins per group and D > 256 the channel chunks."""
    hip = _hip()
    rs = np.random.RandomState(19)
    S = 4
    C = D * G * G
    data = rs.standard_normal((B, C, H, W)).astype(np.float32)
    rois = np.zeros((R, 5), np.float32)
    rois[:, 0] = rs.randint(0, B, R)
    c = rs.uniform(20, SC * min(H, W) - 20, (R, 2))
    wh = rs.uniform(4, 120, (R, 2))
    rois[:, 1:3], rois[:, 3:5] = c - wh / 2, c + wh / 2
    rois[0, 1:] = [-30, -20, 40, 50]
    rois[1, 1:] = [0, 0, SC * W - 1, SC * H - 1]
    rois[2, 1:] = [33, 47, 34, 48]
    trans = (rs.standard_normal((R, 2, P, P)) * 0.5).astype(np.float32)
    dd = torch.from_numpy(np.ascontiguousarray(data.transpose(0, 2, 3, 1))).to(dev()).half()
    td = lambda z: torch.from_numpy(z).to(dev())
    ws = torch.empty(hip.query('sn_dpsroi_bwd_workspace_bytes', R), dtype=torch.uint8, device=dev())
    for tr, tstd in ((None, 0.0), (trans, 0.1)):
        out = torch.empty((R, P, P, D), dtype=torch.float16, device=dev())
        hip.call('sn_psroi_pool_fwd', dd, td(rois), None if tr is None else td(tr), out, R, H, W, D, G, P, S, 1.0 / SC, tstd, 0,
                 hip.stream())
        want = onn.dpsroi_pool(f16r(data).astype(np.float64), rois, tr, P, S, 1.0 / SC, tstd, group_size=G)
        assert_close(out.float().cpu().numpy().transpose(0, 3, 1, 2), want, 1e-2, 1e-2, 'psroi fwd')
        dout = rs.standard_normal((R, D, P, P)).astype(np.float32)
        dod = torch.from_numpy(np.ascontiguousarray(dout.transpose(0, 2, 3, 1))).to(dev()).half()
        wd, wtr = onn.dpsroi_pool_backward(f16r(dout).astype(np.float64), f16r(data).astype(np.float64), rois, tr, P, S, 1.0 / SC,
                                           tstd, group_size=G)
        for f32 in (1, 0):
            d_data = torch.full((B, H, W, C), 7.0, dtype=torch.float32 if f32 else torch.float16, device=dev())
            d_trans = torch.full((R, 2, P, P), 7.0, dtype=torch.float32, device=dev())
            hip.call('sn_psroi_pool_bwd', dod, dd, td(rois), None if tr is None else td(tr), d_data, f32,
                     d_trans if tr is not None else None, R, B, H, W, D, G, P, S, 1.0 / SC, tstd, 0, ws, hip.stream())
            tol = 1e-3 if f32 else 1e-2
            assert_close(d_data.float().cpu().numpy().transpose(0, 3, 1, 2), wd, tol, tol * np.abs(wd).max(), 'psroi d_data')
            if tr is not None:
                assert_close(d_trans.cpu().numpy(), wtr, 1e-3, 1e-3 * np.abs(wtr).max(), 'psroi d_trans')
        d2 = torch.empty_like(d_data)
        hip.call('sn_psroi_pool_bwd', dod, dd, td(rois), None if tr is None else td(tr), d2, 0,
                 d_trans if tr is not None else None, R, B, H, W, D, G, P, S, 1.0 / SC, tstd, 0, ws, hip.stream())
        assert torch.equal(d2, d_data)
        # group-major layout (round 6): the same map with its channels in (gh, gw, d) order -> the same pooled values (the forward's
        # arithmetic per output does not depend on the layout: bit-equal), the data gradient in that order (its summation order over
        # the (RoI, bin) entries is the list order in both kernels: bit-equal too), the same offset gradient
        perm = np.array([d * G * G + g for g in range(G * G) for d in range(D)])
        dgm = dd[..., torch.from_numpy(perm).to(dev())].contiguous()
        out_gm = torch.full_like(out, 7.0)
        hip.call('sn_psroi_pool_fwd', dgm, td(rois), None if tr is None else td(tr), out_gm, R, H, W, D, G, P, S, 1.0 / SC, tstd, 1,
                 hip.stream())
        assert torch.equal(out_gm, out)
        d_gm = torch.full((B, H, W, C), 7.0, dtype=torch.float16, device=dev())
        dt_gm = torch.full((R, 2, P, P), 7.0, dtype=torch.float32, device=dev())
        hip.call('sn_psroi_pool_bwd', dod, dgm, td(rois), None if tr is None else td(tr), d_gm, 0,
                 dt_gm if tr is not None else None, R, B, H, W, D, G, P, S, 1.0 / SC, tstd, 1, ws, hip.stream())
        assert torch.equal(d_gm, d_data[..., torch.from_numpy(perm).to(dev())])
        if tr is not None:
            assert torch.equal(dt_gm, d_trans)


def test_global_average_pool():
    hip = _hip()
    rs = np.random.RandomState(23)
    N, HW, C = 37, 49, 81
    x = rs.standard_normal((N, HW, C)).astype(np.float32)
    xd = torch.from_numpy(x).to(dev()).half()
    y = torch.empty((N, C), dtype=torch.float32, device=dev())
    hip.call('sn_avgpool_global_fwd', xd, y, N, HW, C, hip.stream())
    assert_close(y.cpu().numpy(), f16r(x).mean(1), 1e-5, 1e-5, 'global avg pool')
    dy = rs.standard_normal((N, C)).astype(np.float32)
    dx = torch.full((N, HW, C), 7.0, dtype=torch.float16, device=dev())
    hip.call('sn_avgpool_global_bwd', torch.from_numpy(dy).to(dev()), dx, N, HW, C, hip.stream())
    assert_close(dx.float().cpu().numpy(), np.broadcast_to(dy[:, None, :] / HW, (N, HW, C)), 1e-3, 1e-3, 'global avg pool bwd')


@pytest.mark.parametrize('N,C,H,W,DG', [(2, 64, 7, 6, 4), (2, 512, 9, 11, 4), (1, 64, 5, 5, 1)])
def test_deformable_sampling_vs_oracle(N, C, H, W, DG):
    hip = _hip()
    rs = np.random.RandomState(10)
    KH = KW = 3
    T = 9
    data = rs.standard_normal((N, C, H, W)).astype(np.float32)
    off = (rs.standard_normal((N, 2 * T * DG, H, W)) * 1.5).astype(np.float32)
    off[0, :, 0, 0] = 40.0   # far outside: contributes nothing
    dd = to_nhwc_f16(data)
    offd = torch.from_numpy(np.ascontiguousarray(off.transpose(0, 2, 3, 1))).to(dev())
    col = torch.empty((N * H * W, T, C), dtype=torch.float16, device=dev())
    hip.call('sn_deform_im2col', dd, offd, col, N, H, W, C, KH, KW, 1, 2, 2, DG, 2 * T * DG, 1, hip.stream())
    want = onn.deform_im2col(f16r(data).astype(np.float64), off.astype(np.float64), KH, KW, 1, 2, 2, DG)
    assert_close(col.float().cpu().numpy().reshape(N, H, W, T, C), want, 1e-2, 1e-2, 'deform im2col')
    # zero offsets == plain dilated im2col
    col0 = torch.empty_like(col)
    hip.call('sn_deform_im2col', dd, torch.zeros_like(offd).half(), col0, N, H, W, C, KH, KW, 1, 2, 2, DG, 2 * T * DG, 0, hip.stream())
    unf = Fnn.unfold(torch.from_numpy(f16r(data)), 3, dilation=2, padding=2).numpy().reshape(N, C, T, H, W)
    assert_close(col0.float().cpu().numpy().reshape(N, H, W, T, C), unf.transpose(0, 3, 4, 2, 1), 1e-3, 1e-3, 'deform zero offset')
    dcol = rs.standard_normal((N, H, W, T, C)).astype(np.float32)
    wdata, woff = onn.deform_col2im(f16r(dcol).astype(np.float64), f16r(data).astype(np.float64), off.astype(np.float64), KH, KW, 1, 2, 2, DG)
    dcd = torch.from_numpy(dcol).to(dev()).half()
    for f32 in (1, 0):
        d_data = torch.full((N, H, W, C), 7.0, dtype=torch.float32 if f32 else torch.float16, device=dev())
        d_off = torch.full((N, H, W, 2 * T * DG), 7.0, dtype=torch.float32, device=dev())
        hip.call('sn_deform_col2im', dcd, dd, offd, d_data, f32, d_off, N, H, W, C, KH, KW, 1, 2, 2, DG, 2 * T * DG, 1, None, hip.stream())
        # the pruned candidate scan (max |offset| window) visits the same samples in the same order: bit-identical
        d_data2 = torch.full_like(d_data, 7.0)
        wsd = torch.zeros(16, dtype=torch.uint8, device=dev())
        hip.call('sn_deform_col2im', dcd, dd, offd, d_data2, f32, None, N, H, W, C, KH, KW, 1, 2, 2, DG, 2 * T * DG, 1, wsd, hip.stream())
        assert torch.equal(d_data2, d_data)
        assert wsd.view(torch.float32)[0].item() == float(np.abs(off).max())
        # sub-cell offsets: the window really prunes (the 40-cell outlier above opens it completely)
        small = torch.from_numpy(np.ascontiguousarray((off * 0.2).clip(-0.9, 0.9).transpose(0, 2, 3, 1))).to(dev())
        a, b = torch.full_like(d_data, 7.0), torch.full_like(d_data, 7.0)
        hip.call('sn_deform_col2im', dcd, dd, small, a, f32, None, N, H, W, C, KH, KW, 1, 2, 2, DG, 2 * T * DG, 1, None, hip.stream())
        hip.call('sn_deform_col2im', dcd, dd, small, b, f32, None, N, H, W, C, KH, KW, 1, 2, 2, DG, 2 * T * DG, 1, wsd, hip.stream())
        assert torch.equal(a, b) and float(a.float().abs().sum()) > 0
        tol = 1e-3 if f32 else 1e-2
        assert_close(d_data.float().cpu().numpy().transpose(0, 3, 1, 2), wdata, tol, tol * np.abs(wdata).max(), 'deform d_data')
        assert_close(d_off.cpu().numpy().transpose(0, 3, 1, 2), woff, 1e-3, 1e-3 * np.abs(woff).max(), 'deform d_offset')


@pytest.mark.parametrize('N,C,H,W,s', [(2, 32, 12, 10, 1), (2, 96, 11, 13, 2), (1, 960, 8, 8, 1), (3, 1024 + 64, 5, 6, 2)])
def test_depthwise_conv_vs_torch(N, C, H, W, s):
    """MobileNetV2 depthwise 3x3 (mobilenetv2_e2e.py:57-66): forward, data gradient (+accumulate), weight gradient."""
    hip = _hip()
    rs = np.random.RandomState(C + s)
    x = rs.standard_normal((N, C, H, W)).astype(np.float32)
    w = (rs.standard_normal((C, 1, 3, 3)) / 3).astype(np.float32)
    xt, wt = torch.from_numpy(f16r(x)).requires_grad_(True), torch.from_numpy(f16r(w)).requires_grad_(True)
    y = Fnn.conv2d(xt, wt, None, s, 1, 1, groups=C)
    dy = rs.standard_normal(tuple(y.shape)).astype(np.float32)
    y.backward(torch.from_numpy(f16r(dy)))
    Ho, Wo = y.shape[2], y.shape[3]
    xd, wd = to_nhwc_f16(x), torch.from_numpy(w.reshape(C, 9)).to(dev()).half().contiguous()
    yd = torch.empty((N, Ho, Wo, C), dtype=torch.float16, device=dev())
    hip.call('sn_dwconv_fwd', xd, wd, yd, N, H, W, C, C, C, 3, 3, s, 1, 1, hip.stream())
    assert_close(from_nhwc(yd), y.detach().numpy(), 1e-2, 1e-2 * np.abs(y.detach().numpy()).max(), 'dw fwd')
    dyd = to_nhwc_f16(dy)
    dx = torch.full((N, H, W, C), 7.0, dtype=torch.float16, device=dev())
    hip.call('sn_dwconv_dgrad', dyd, wd, None, dx, N, H, W, C, C, C, C, 3, 3, s, 1, 1, hip.stream())
    want_dx = xt.grad.numpy()
    assert_close(from_nhwc(dx), want_dx, 1e-2, 1e-2 * np.abs(want_dx).max(), 'dw dgrad')
    hip.call('sn_dwconv_dgrad', dyd, wd, dx, dx, N, H, W, C, C, C, C, 3, 3, s, 1, 1, hip.stream())
    assert_close(from_nhwc(dx), 2 * want_dx, 2e-2, 2e-2 * np.abs(want_dx).max(), 'dw dgrad accumulate')
    need = hip.query('sn_dwconv_wgrad_workspace_bytes', N, H, W, C, 3, 3, s, 1, 1)
    ws = torch.empty(need, dtype=torch.uint8, device=dev())
    runs = []
    for rep in range(3):         # per-block partials summed in block order: the same bits every run
        dw = torch.zeros((C, 9), dtype=torch.float32, device=dev())
        hip.call('sn_dwconv_wgrad', dyd, xd, dw, N, H, W, C, C, C, 3, 3, s, 1, 1, ws, need, hip.stream())
        runs.append(dw.clone())
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
    want_dw = wt.grad.numpy().reshape(C, 9)
    assert_close(dw.cpu().numpy(), want_dw, 1e-2, 1e-2 * np.abs(want_dw).max(), 'dw wgrad')


def test_clip_and_bn_relu6():
    hip = _hip()
    rs = np.random.RandomState(21)
    N, C, H, W = 2, 24, 6, 5
    M = N * H * W
    x = (rs.standard_normal((N, C, H, W)) * 4 + 2).astype(np.float32)
    dy = rs.standard_normal((N, C, H, W)).astype(np.float32)
    xd, dyd = to_nhwc_f16(x), to_nhwc_f16(dy)
    y = torch.empty_like(xd)
    hip.call('sn_clip_f16', xd, None, None, y, M, C, C, C, C, C, 0.0, 6.0, 0, hip.stream())
    assert_close(from_nhwc(y), np.clip(f16r(x), 0, 6), 1e-3, 1e-3, 'clip fwd')
    dx = torch.empty_like(xd)
    hip.call('sn_clip_f16', dyd, xd, None, dx, M, C, C, C, C, C, 0.0, 6.0, 1, hip.stream())
    xr = f16r(x)
    assert_close(from_nhwc(dx), np.where((xr >= 0) & (xr <= 6), f16r(dy), 0), 1e-3, 1e-3, 'clip bwd')
    # BatchNorm + relu6 fused (activation code 2) against torch: y = clip(BN(x), 0, 6)
    gamma, beta = rs.uniform(1.0, 3.0, C).astype(np.float32), rs.uniform(0, 3, C).astype(np.float32)
    f = lambda n: torch.zeros(n, dtype=torch.float32, device=dev())
    ws = torch.empty(hip.query('sn_bn_workspace_bytes', M, C), dtype=torch.uint8, device=dev())
    scale, shift, mean, invstd, rm, rv = f(C), f(C), f(C), f(C), f(C), f(C)
    g_d, b_d = torch.from_numpy(gamma).to(dev()), torch.from_numpy(beta).to(dev())
    hip.call('sn_bn_stats', xd, M, C, C, ws, hip.stream())
    hip.call('sn_bn_finalize', ws, M, C, 1e-5, 0.9, g_d, b_d, rm, rv, scale, shift, mean, invstd, hip.stream())
    hip.call('sn_bn_apply', xd, y, M, C, C, C, scale, shift, 2, hip.stream())
    xt = torch.from_numpy(xr).requires_grad_(True)
    gt, bt = torch.from_numpy(gamma).requires_grad_(True), torch.from_numpy(beta).requires_grad_(True)
    yt = torch.clamp(Fnn.batch_norm(xt, None, None, gt, bt, True, 0.0, 1e-5), 0, 6)
    yt.backward(torch.from_numpy(f16r(dy)))
    assert_close(from_nhwc(y), yt.detach().numpy(), 1e-2, 1e-2 * float(yt.detach().abs().max()), 'bn relu6 fwd')
    dg, db = f(C), f(C)
    hip.call('sn_bn_backward', dyd, xd, None, dx, M, C, C, C, C, C, scale, shift, mean, invstd, 2, ws, dg, db, hip.stream())
    assert_close(db.cpu().numpy(), bt.grad.numpy(), 3e-2, 3e-2 * np.abs(bt.grad.numpy()).max(), 'bn relu6 dbeta')
    assert_close(dg.cpu().numpy(), gt.grad.numpy(), 3e-2, 3e-2 * np.abs(gt.grad.numpy()).max(), 'bn relu6 dgamma')
    assert_close(from_nhwc(dx), xt.grad.numpy(), 3e-2, 3e-2 * np.abs(xt.grad.numpy()).max(), 'bn relu6 dx')


def test_stem_conv_3x3_wgrad_packed():
    """MobileNetV2's first convolution (3 -> 32, 3x3/2, pad 1; mobilenetv2_e2e.py:196-204) is trainable: forward and
    weight gradient on the packed NHWC4 input, weight in the packed [O][KH][KWP*4] layout."""
    hip = _hip()
    rs = np.random.RandomState(31)
    N, H, W, O, K, s, pad = 2, 32, 40, 32, 3, 2, 1
    x = (rs.standard_normal((N, 3, H, W)) * 2).astype(np.float32)
    w = (rs.standard_normal((O, 3, K, K)) / 5).astype(np.float32)
    Ho, Wo = (H + 2 * pad - K) // s + 1, (W + 2 * pad - K) // s + 1
    KWP = (K + 1) // 2 * 2
    Hp, Wp = (Ho - 1) * s + K, ((Wo - 1) * s + KWP + 1) // 2 * 2
    xp = torch.empty((N, Hp, Wp, 4), dtype=torch.float16, device=dev())
    hip.call('sn_pack_stem_input', torch.from_numpy(x).to(dev()), xp, N, 3, H, W, Hp, Wp, pad, pad, None, None, hip.stream())
    wk = np.zeros((O, K, KWP, 4), np.float32)
    wk[:, :, :K, :3] = w.transpose(0, 2, 3, 1)
    wd = torch.from_numpy(wk.reshape(O, K, KWP * 4)).to(dev()).half().contiguous()
    y = torch.empty((N, Ho, Wo, O), dtype=torch.float16, device=dev())
    hip.call('sn_conv_stem_fwd', xp, wd, None, y, N, Hp, Wp, Ho, Wo, O, O, K, KWP, s, 0, 0, hip.stream())
    xt, wt = torch.from_numpy(f16r(x)), torch.from_numpy(f16r(w)).requires_grad_(True)
    yt = Fnn.conv2d(xt, wt, None, s, pad)
    assert_close(from_nhwc(y), yt.detach().numpy(), 1e-2, 1e-2 * float(yt.abs().max()), 'stem 3x3 fwd')
    dy = rs.standard_normal(tuple(yt.shape)).astype(np.float32)
    yt.backward(torch.from_numpy(f16r(dy)))
    dw = torch.zeros((O, K, KWP * 4), dtype=torch.float32, device=dev())
    need = hip.query('sn_conv_stem_wgrad_workspace_bytes', N, Ho, Wo, O, K, KWP)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dw.device)
    hip.call('sn_conv_stem_wgrad', to_nhwc_f16(dy), xp, dw, N, Hp, Wp, Ho, Wo, O, O, K, KWP, s, ws, need, hip.stream())
    got = dw.cpu().numpy().reshape(O, K, KWP, 4)
    want = wt.grad.numpy().transpose(0, 2, 3, 1)
    assert_close(got[:, :, :K, :3], want, 1e-2, 1e-2 * np.abs(want).max(), 'stem wgrad')
    assert float(np.abs(got[:, :, :, 3]).max()) == 0.0            # channel padding carries no gradient
