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
    # the slab-stationary launch (what the executor calls: B known) writes the same rows, bit for bit; unsorted RoIs too
    out_s = torch.full((R, P, P, C), 7.0, dtype=torch.float16, device=dev())
    hip.call('sn_dpsroi_pool_fwd_images', dd, td(rois), None if trans is None else td(trans), out_s, R, B, H, W, C, P, S, 1.0 / SC, tstd,
             hip.stream())
    assert torch.equal(out_s, out)
    perm = rs.permutation(R)
    hip.call('sn_dpsroi_pool_fwd_images', dd, td(rois[perm]), None if trans is None else td(trans[perm]), out_s, R, B, H, W, C, P, S,
             1.0 / SC, tstd, hip.stream())
    assert torch.equal(out_s, out[torch.from_numpy(perm).to(dev())])
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


@pytest.mark.parametrize('D', [81, 4])
def test_position_sensitive_pool_at_c4_launch_shape(D):
    """BASELINE configs[3] (C4, R-FCN head) at one GPU's share: 16 chips x 300 RoIs on the 32 x 32 maps of 7 * 7 * D channels (D = 81:
    rfcn_cls, 3969 channels -> sixteen 256-channel chunks in the data gradient; D = 4: rfcn_bbox), group_size 7, pooled offsets --
    sn_psroi_pool_fwd / _bwd against oracle/nn.py (sparse-operator form; position-sensitive variant spec ours, SURVEY 8(d) C4;
    resnet_mx_101_e2e.py:286-293 is the group_size = 1 call it replaces).  The kernels run the full launch shape; the CPU oracle is
    evaluated on every fourth RoI: forward rows of that subset, and the backward's output gradient is zero outside it (the data gradient
    is linear in it, the other RoIs' windows are still scanned), which keeps the float64 operator products to seconds."""
    from sniper_amd import hip
    rs = np.random.RandomState(41 + D)
    Bc, G, H, W, P, S, SC = 16, 7, 32, 32, 7, 4, 16
    R, C = Bc * 300, D * G * G
    data = rs.standard_normal((Bc, C, H, W)).astype(np.float32)
    rois = np.zeros((R, 5), np.float32)
    rois[:, 0] = np.repeat(np.arange(Bc), 300)
    c = rs.uniform(0, 512, (R, 2))
    wh = np.exp(rs.uniform(np.log(8), np.log(400), (R, 2)))
    rois[:, 1:3], rois[:, 3:5] = np.clip(c - wh / 2, 0, 511), np.clip(c + wh / 2, 0, 511)
    trans = (rs.standard_normal((R, 2, P, P)) * 0.5).astype(np.float32)
    tstd = 0.1
    sel = np.sort(rs.permutation(R)[:R // 4])
    dd = torch.from_numpy(np.ascontiguousarray(data.transpose(0, 2, 3, 1))).to(dev()).half()
    td = lambda z: torch.from_numpy(z).to(dev())
    out = torch.full((R, P, P, D), 7.0, dtype=torch.float16, device=dev())
    # (group-major maps, as the executor lays them out: channel (gh*G + gw)*D + d of the device tensor = operator channel (d*G + gh)*G + gw)
    perm = torch.from_numpy(np.array([d * G * G + g for g in range(G * G) for d in range(D)])).to(dev())
    dd = dd[..., perm].contiguous()
    hip.call('sn_psroi_pool_fwd', dd, td(rois), td(trans), out, R, H, W, D, G, P, S, 1.0 / SC, tstd, 1, hip.stream())
    want = onn.dpsroi_pool_fast(f16r(data), rois[sel], trans[sel], P, S, 1.0 / SC, tstd, group_size=G)
    got = out.float().cpu().numpy().transpose(0, 3, 1, 2)
    assert_close(got[sel], want, 1e-2, 1e-2, 'psroi fwd C4 D=%d' % D)
    assert np.isfinite(got).all() and not (got == 7.0).all(axis=(1, 2, 3)).any()       # every RoI's row was written
    dout = np.zeros((R, D, P, P), np.float32)
    dout[sel] = rs.standard_normal((len(sel), D, P, P))
    dod = torch.from_numpy(np.ascontiguousarray(dout.transpose(0, 2, 3, 1))).to(dev()).half()
    wd, wtr = onn.dpsroi_pool_backward_fast(f16r(dout[sel]), f16r(data), rois[sel], trans[sel], P, S, 1.0 / SC, tstd, group_size=G)
    ws = torch.empty(hip.query('sn_dpsroi_bwd_workspace_bytes', R), dtype=torch.uint8, device=dev())
    d_data = torch.full((Bc, H, W, C), 7.0, dtype=torch.float16, device=dev())
    d_trans = torch.full((R, 2, P, P), 7.0, dtype=torch.float32, device=dev())
    hip.call('sn_psroi_pool_bwd', dod, dd, td(rois), td(trans), d_data, 0, d_trans, R, Bc, H, W, D, G, P, S, 1.0 / SC, tstd, 1, ws,
             hip.stream())
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(C, device=dev())
    got_d = d_data[..., inv].float().cpu().numpy().transpose(0, 3, 1, 2)       # back to the operator's channel order
    assert_close(got_d, wd, 1e-2, 1e-2 * np.abs(wd).max(), 'psroi d_data C4 D=%d' % D)
    gtr = d_trans.cpu().numpy()
    assert_close(gtr[sel], wtr, 1e-2, 1e-2 * np.abs(wtr).max(), 'psroi d_trans C4 D=%d' % D)
    rest = np.setdiff1d(np.arange(R), sel)
    assert not gtr[rest].any()                                                         # zero output gradient -> zero offset gradient
    d2 = torch.empty_like(d_data)
    hip.call('sn_psroi_pool_bwd', dod, dd, td(rois), td(trans), d2, 0, d_trans, R, Bc, H, W, D, G, P, S, 1.0 / SC, tstd, 1, ws, hip.stream())
    assert torch.equal(d2, d_data)                                                     # fixed summation order


def _fused_case(name):
    N, C, H, W, O, K, s, p, d, hb, hr, relu = C2_CONV_SHAPES[name]
    Ho = (H + 2 * p - d * (K - 1) - 1) // s + 1
    return N, C, H, W, O, K, s, p, d, hb, hr, relu, Ho, Ho


@pytest.mark.parametrize('name', list(C2_CONV_SHAPES))
def test_conv_fwd_stats_at_c2_launch_shapes_vs_oracle(name):
    """sn_conv_fwd_stats (87 of the step's 115 forward launches) at the C2 launch shapes against the fp32 oracle DIRECTLY:
    the output tensor (1e-2) and the BatchNorm batch statistics its epilogue emits -- mean and biased variance per channel
    from the per-row-tile partials against mean / variance of the oracle's tensor (mean to 1e-2 of the channel's standard
    deviation, variance to 1e-2 relative)."""
    from sniper_amd import hip
    from gpu_util import from_nhwc, to_nhwc_f16, w_to_otI
    N, C, H, W, O, K, s, p, d, hb, hr, relu, Ho, Wo = _fused_case(name)
    args = (N, H, W, C, C, O, O, O if hr else 0, K, K, s, p, d)
    nblk = hip.query('sn_conv_fwd_stats_blocks', *args)
    if nblk <= 0:
        # the engine asks the same query and falls back to sn_conv_fwd + sn_bn_stats for such a layer (Cout = 72: narrow)
        assert O < 128
        pytest.skip('%s does not qualify for the statistics epilogue (Cout = %d)' % (name, O))
    rs = np.random.RandomState(len(name) * 11 + O)
    x = rs.standard_normal((N, C, H, W)).astype(np.float32)
    w = (rs.standard_normal((O, C, K, K)) / np.sqrt(C * K * K)).astype(np.float32)
    b = rs.standard_normal(O).astype(np.float32) if hb else None
    res = rs.standard_normal((N, O, Ho, Wo)).astype(np.float32) if hr else None
    want = _ref_conv(x, w, b, res, s, p, d, relu)
    xd = to_nhwc_f16(x)
    wd = torch.from_numpy(w_to_otI(w)).to(dev()).half().contiguous()
    bd = torch.from_numpy(b).to(dev()) if hb else None
    rd = to_nhwc_f16(res) if hr else None
    y = torch.full((N, Ho, Wo, O), 7.0, dtype=torch.float16, device=dev())
    part = torch.full((nblk, 2, O), 7.0, dtype=torch.float32, device=dev())
    hip.call('sn_conv_fwd_stats', xd, wd, bd, rd, y, *args, relu, part, hip.stream())
    torch.cuda.synchronize()
    assert_close(from_nhwc(y), want, 1e-2, 1e-2 * np.abs(want).max(), 'conv fwd (+stats) %s' % name)
    M = N * Ho * Wo
    sm, sq = part[:, 0].double().sum(0).cpu().numpy(), part[:, 1].double().sum(0).cpu().numpy()
    mean, var = sm / M, sq / M - (sm / M) ** 2
    w64 = want.astype(np.float64).transpose(1, 0, 2, 3).reshape(O, -1)
    wmean, wvar = w64.mean(1), w64.var(1)
    assert np.all(np.abs(mean - wmean) <= 1e-2 * np.sqrt(wvar)), float(np.max(np.abs(mean - wmean) / np.sqrt(wvar)))
    assert np.all(np.abs(var - wvar) <= 1e-2 * wvar), float(np.max(np.abs(var - wvar) / wvar))


# N, C, H, W, O, residual, relu: 1x1 layers of the 20-chip step whose tile count divides over the 512 resident workgroups
PERSIST_SHAPES = {
    'stage3 256->1024 @32 (+residual)': (B, 256, 32, 32, 1024, True, 0),
    'stage2 128->512 @64 (+residual)': (B, 128, 64, 64, 512, True, 0),
    'stage4 512->2048 @32': (B, 512, 32, 32, 2048, False, 1),
}


@pytest.mark.parametrize('name', list(PERSIST_SHAPES))
def test_persistent_tile_loop_equals_one_tile_per_workgroup(name):
    """Round 6: launches of >= 4 tiles per CU run as 512 persistent workgroups that walk their tiles as one pipeline (conv_dma.hip,
    configurations 24 / 26).  Same arithmetic per output, so everything must be BIT-equal to the one-tile-per-workgroup launch
    (sn_debug_option conv_no_persist): the forward with residual and the statistics partials of its epilogue, the plain data
    gradient, and the data gradient with the BatchNorm-backward reduction (coefficients through LDS-DMA in the persistent kernel).
    The comparison against the oracle at these shapes is the tests above (they run the persistent default)."""
    from sniper_amd import hip
    from gpu_util import to_nhwc_f16, w_to_otI
    N, C, H, W, O, hr, relu = PERSIST_SHAPES[name]
    rs = np.random.RandomState(len(name) + C)
    x = to_nhwc_f16(rs.standard_normal((N, C, H, W)).astype(np.float32))
    w = (rs.standard_normal((O, C, 1, 1)) / np.sqrt(C)).astype(np.float32)
    wd = torch.from_numpy(w_to_otI(w)).to(dev()).half().contiguous()
    res = to_nhwc_f16(rs.standard_normal((N, O, H, W)).astype(np.float32)) if hr else None
    args = (N, H, W, C, C, O, O, O if hr else 0, 1, 1, 1, 0, 1)
    nblk = hip.query('sn_conv_fwd_stats_blocks', *args)
    assert nblk > 0
    # backward operands: dy (N, H, W, O) -> dx (N, H, W, C) needs a persistent-sized launch too: use the transposed roles, i.e. the
    # data gradient of the REDUCTION O -> C' = the forward's input width seen from the other side (dy has C channels, dx has O)
    dy = to_nhwc_f16(rs.standard_normal((N, C, H, W)).astype(np.float32))
    w2 = (rs.standard_normal((C, O, 1, 1)) / np.sqrt(O)).astype(np.float32)               # a conv O -> C; its dgrad: dy (.., C) -> dx (.., O)
    wT = torch.empty((O, 1, C), dtype=torch.float16, device=dev())
    hip.call('sn_weight_transpose', torch.from_numpy(w_to_otI(w2)).to(dev()), wT, C, 1, O, C, hip.stream())
    dargs = (N, H, W, O, O, C, C, 0, 1, 1, 1, 0, 1)
    dblk = hip.query('sn_conv_dgrad_bn_blocks', *dargs)
    assert dblk > 0
    bnx = to_nhwc_f16(rs.standard_normal((N, O, H, W)).astype(np.float32))
    f = lambda a: torch.from_numpy(np.asarray(a, np.float32)).to(dev())
    dsc, dsh, dmu = f(rs.uniform(0.5, 1.5, O)), f(rs.uniform(-0.3, 0.6, O)), f(rs.standard_normal(O) * 0.1)
    acc = to_nhwc_f16(rs.standard_normal((N, O, H, W)).astype(np.float32))
    out = []
    for no_persist in (1, 0):
        hip.call('sn_debug_option', b'conv_no_persist', no_persist)
        try:
            y = torch.full((N, H, W, O), 7.0, dtype=torch.float16, device=dev())
            part = torch.full((nblk, 2, O), 7.0, dtype=torch.float32, device=dev())
            hip.call('sn_conv_fwd_stats', x, wd, None, res, y, *args, relu, part, hip.stream())
            y2 = torch.full((N, H, W, O), 7.0, dtype=torch.float16, device=dev())
            hip.call('sn_conv_fwd', x, wd, None, res, y2, N, H, W, C, C, O, O, O if hr else 0, 1, 1, 1, 0, 1, relu, 0, hip.stream())
            dx = torch.full((N, H, W, O), 7.0, dtype=torch.float16, device=dev())
            bpart = torch.full((dblk, 2, O), 7.0, dtype=torch.float32, device=dev())
            hip.call('sn_conv_dgrad_bn', dy, wT, None, dx, *dargs, bnx, O, dsc, dsh, dmu, 1, bpart, hip.stream())
            dx2 = torch.full((N, H, W, O), 7.0, dtype=torch.float16, device=dev())
            hip.call('sn_conv_dgrad', dy, wT, acc, dx2, N, H, W, O, O, C, C, O, 1, 1, 1, 0, 1, 0, hip.stream())
            torch.cuda.synchronize()
            out.append((y, part, y2, dx, bpart, dx2))
        finally:
            hip.call('sn_debug_option', b'conv_no_persist', 0)
    for k, (a, b) in enumerate(zip(*out)):
        assert torch.equal(a, b), ('fwd+stats y', 'fwd statistics partials', 'fwd y', 'dgrad+bn dx', 'BatchNorm-backward partials', 'dgrad (+accumulate) dx')[k]
    assert torch.isfinite(out[1][0].float()).all() and not (out[1][0] == 7.0).all()


@pytest.mark.parametrize('name,act', [(n, 1) for n in C2_CONV_SHAPES] + [('stage3 1x1 1024->256 @32', 0)])
def test_conv_dgrad_bn_at_c2_launch_shapes_vs_oracle(name, act):
    """sn_conv_dgrad_bn (84 of the step's data-gradient launches) + sn_bn_backward_blocks at the C2 launch shapes against the
    fp32 oracle DIRECTLY: the data gradient (torch-CPU autograd of the fp32 convolution), and -- from the partial sums its
    epilogue emits -- dgamma, dbeta and the input gradient of the BatchNorm(+ReLU) below it against the numpy statement of the
    BatchNorm backward on the ORACLE's data gradient.  1e-2 of each tensor's scale."""
    import torch.nn.functional as Fnn
    from sniper_amd import hip
    from gpu_util import from_nhwc, to_nhwc_f16, w_to_otI
    N, C, H, W, O, K, s, p, d, hb, hr, relu, Ho, Wo = _fused_case(name)
    Op = (O + 7) // 8 * 8
    args = (N, H, W, C, C, Op, Op, 0, K, K, s, p, d)
    nblk = hip.query('sn_conv_dgrad_bn_blocks', *args)
    if nblk <= 0:
        pytest.skip('%s does not qualify for the BatchNorm-backward epilogue' % name)
    rs = np.random.RandomState(len(name) * 13 + C + act)
    w = (rs.standard_normal((O, C, K, K)) / np.sqrt(C * K * K)).astype(np.float32)
    dy = rs.standard_normal((N, O, Ho, Wo)).astype(np.float32)
    bnx = rs.standard_normal((N, C, H, W)).astype(np.float32)
    gamma, beta = rs.uniform(0.5, 1.5, C), rs.uniform(-0.3, 0.6, C)
    # the saved statistics are inputs of the entry (any values: the backward formula is algebra in them); float32-exact
    mean = (rs.standard_normal(C) * 0.1).astype(np.float32).astype(np.float64)
    invstd = rs.uniform(0.7, 1.6, C).astype(np.float32).astype(np.float64)
    scale = (gamma * invstd).astype(np.float32).astype(np.float64)
    shift = (beta - mean * gamma * invstd).astype(np.float32).astype(np.float64)
    bc = lambda v: v.reshape(1, C, 1, 1)
    # keep every pre-activation away from the ReLU threshold: a mask decided by the last bit of x * scale + shift is not an
    # arithmetic difference this test is about (the end-to-end runs teacher-force it for the same reason)
    bnx = f16r(bnx)
    near = np.abs(bnx.astype(np.float64) * bc(scale) + bc(shift)) < 1e-2
    bnx = f16r(bnx + 0.125 * near)
    x16 = bnx.astype(np.float64)
    assert not (np.abs(x16 * bc(scale) + bc(shift)) < 1e-3).any()
    # oracle: dL/d(conv input) by autograd of the fp32 convolution (its input value does not matter: the op is linear)
    xt = torch.zeros((N, C, H, W), requires_grad=True)
    Fnn.conv2d(xt, torch.from_numpy(f16r(w)), None, s, p, d).backward(torch.from_numpy(f16r(dy)))
    want_dx = xt.grad.numpy().astype(np.float64)
    z = x16 * bc(scale) + bc(shift)
    g = want_dx * (z > 0) if act == 1 else want_dx
    M = N * H * W
    want_dbeta = g.sum((0, 2, 3))
    gx = (g * (x16 - bc(mean))).sum((0, 2, 3))
    want_dgamma = gx * invstd
    want_dbn = bc(scale) * (g - bc(want_dbeta) / M - (x16 - bc(mean)) * bc(invstd ** 2 * gx) / M)
    d_dy = torch.zeros((N, Ho, Wo, Op), dtype=torch.float16, device=dev())
    d_dy[..., :O] = to_nhwc_f16(dy)
    wT = torch.empty((C, K * K, Op), dtype=torch.float16, device=dev())
    hip.call('sn_weight_transpose', torch.from_numpy(w_to_otI(w)).to(dev()), wT, O, K * K, C, Op, hip.stream())
    f = lambda a: torch.from_numpy(np.asarray(a, np.float32)).to(dev())
    dsc, dsh, dmean, dinv = f(scale), f(shift), f(mean), f(invstd)
    bnxd = to_nhwc_f16(bnx)
    dx = torch.full((N, H, W, C), 7.0, dtype=torch.float16, device=dev())
    part = torch.full((nblk, 2, C), 7.0, dtype=torch.float32, device=dev())
    hip.call('sn_conv_dgrad_bn', d_dy, wT, None, dx, *args, bnxd, C, dsc, dsh, dmean, act, part, hip.stream())
    torch.cuda.synchronize()
    assert_close(from_nhwc(dx), want_dx, 1e-2, 1e-2 * np.abs(want_dx).max(), 'dgrad (+bn reduction) %s' % name)
    ws = torch.empty(hip.query('sn_bn_workspace_bytes', M, C), dtype=torch.uint8, device=dev())
    dg, db = torch.zeros(C, device=dev()), torch.zeros(C, device=dev())
    out = torch.full((N, H, W, C), 7.0, dtype=torch.float16, device=dev())
    hip.call('sn_bn_backward_blocks', part, nblk, dx, bnxd, None, out, M, C, C, C, C, C, dsc, dsh, dmean, dinv, act, ws, dg, db,
             hip.stream())
    torch.cuda.synchronize()
    assert_close(db.cpu().numpy(), want_dbeta, 1e-2, 1e-2 * np.abs(want_dbeta).max(), 'dbeta %s' % name)
    assert_close(dg.cpu().numpy(), want_dgamma, 1e-2, 1e-2 * np.abs(want_dgamma).max(), 'dgamma %s' % name)
    assert_close(from_nhwc(out), want_dbn, 1e-2, 1e-2 * np.abs(want_dbn).max(), 'BatchNorm input gradient %s' % name)
