"""GPU: the mask branch (symbols/faster/resnet_mx_101_e2e_mask.py) -- device operators against oracle/nn.py / torch, a training
step through the iterator with WITH_MASK, and the teacher-forced end-to-end comparison with oracle/graph_cpu.py.  The fork's
operators are not in the reference tree: parity unpinned, semantics in DESIGN.md."""
import os

import numpy as np
import pytest
import torch

from gpu_util import assert_close, dev, f16r
from oracle import nn as onn

pytestmark = pytest.mark.gpu


def _hip():
    from sniper_amd import hip
    hip.require_gpu()
    return hip


def test_mask_rcnn_target_bit_exact_vs_oracle():
    """Rasterisation of encoded polygons (incl. truncated objects, several segments, padding RoIs) into 28x28 targets."""
    hip = _hip()
    from sniper_amd.data import mask_utils
    from sniper_amd.synthetic import make_roidb
    rs = np.random.RandomState(4)
    roidb = make_roidb(3, seed=21, with_masks=True)
    B, nm, G, L, ms = len(roidb), 16, 100, 500, 28
    polys = -np.ones((B, G, L), np.float32)
    rois = np.zeros((B * nm, 5), np.float32)
    ids = -np.ones((B * nm,), np.float32)
    for b, r in enumerate(roidb):
        k = len(r['gt_masks'])
        polys[b] = mask_utils.poly_encoder(mask_utils.crop_polys(r['gt_masks'], [0, 0], 1.0), r['gt_classes'][:k] - 1, L, G)
        for j in range(nm):
            rois[b * nm + j, 0] = b
            if j < nm - 3:
                g = int(rs.randint(0, k))
                jit = rs.uniform(-0.2, 0.2, 4) * np.tile(r['boxes'][g, 2:4] - r['boxes'][g, 0:2], 2)
                rois[b * nm + j, 1:] = r['boxes'][g] + jit
                ids[b * nm + j] = g
    td = lambda z: torch.from_numpy(z).to(dev())
    tg = torch.full((B * nm, ms, ms), 7.0, device=dev())
    cl = torch.full((B * nm,), 7.0, device=dev())
    hip.call('sn_mask_rcnn_target', td(rois), td(polys), td(ids), B * nm, nm, G, L, ms, tg, cl, hip.stream())
    want_t, want_c = onn.mask_rcnn_target(rois, polys, ids, nm, ms)
    assert np.array_equal(tg.cpu().numpy(), want_t) and np.array_equal(cl.cpu().numpy(), want_c)
    assert (want_t == 1).sum() > 500 and (want_t == 0).sum() > 500 and (want_t[ids < 0] == -1).all()


def test_deconvolution_pick_and_shuffles():
    hip = _hip()
    rs = np.random.RandomState(5)
    N, H, W, C, O = 3, 5, 4, 16, 24
    x = rs.standard_normal((N, C, H, W)).astype(np.float32)
    w = (rs.standard_normal((C, O, 2, 2)) * 0.3).astype(np.float32)
    xt, wt = torch.from_numpy(f16r(x)).requires_grad_(True), torch.from_numpy(f16r(w)).requires_grad_(True)
    y = torch.nn.functional.conv_transpose2d(xt, wt, None, 2, 0)
    dy = rs.standard_normal(tuple(y.shape)).astype(np.float32)
    y.backward(torch.from_numpy(f16r(dy)))
    xd = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 3, 1))).to(dev()).half()
    wi = torch.from_numpy(np.ascontiguousarray(w.transpose(2, 3, 1, 0)).reshape(4 * O, 1, C)).to(dev()).half()    # rows (a, b, o)
    tmp = torch.empty((N, H, W, 4 * O), dtype=torch.float16, device=dev())
    hip.call('sn_conv_fwd', xd, wi, None, None, tmp, N, H, W, C, C, 4 * O, 4 * O, 0, 1, 1, 1, 0, 1, 0, 0, hip.stream())
    out = torch.full((N, 2 * H, 2 * W, O), 7.0, dtype=torch.float16, device=dev())
    hip.call('sn_depth_to_space2', tmp, out, N, H, W, O, 0, hip.stream())
    assert_close(out.float().cpu().numpy().transpose(0, 3, 1, 2), y.detach().numpy(), 1e-2, 1e-2, 'deconvolution fwd')
    back = torch.full_like(tmp, 7.0)
    hip.call('sn_space_to_depth2', out, back, N, H, W, O, hip.stream())
    assert torch.equal(back, tmp)
    relu = torch.empty_like(out)
    hip.call('sn_depth_to_space2', tmp, relu, N, H, W, O, 1, hip.stream())
    assert torch.equal(relu, torch.relu(out))
    # data gradient = 1x1 data-gradient GEMM of the shuffled dY with the transposed weight
    dyd = torch.from_numpy(np.ascontiguousarray(dy.transpose(0, 2, 3, 1))).to(dev()).half()
    dtmp = torch.empty_like(tmp)
    hip.call('sn_space_to_depth2', dyd, dtmp, N, H, W, O, hip.stream())
    wT = torch.empty((C, 1, 4 * O), dtype=torch.float16, device=dev())
    hip.call('sn_weight_transpose', wi.float(), wT, 4 * O, 1, C, 4 * O, hip.stream())
    dx = torch.empty_like(xd)
    hip.call('sn_conv_dgrad', dtmp, wT, None, dx, N, H, W, C, C, 4 * O, 4 * O, C, 1, 1, 1, 0, 1, 0, hip.stream())
    assert_close(dx.float().cpu().numpy().transpose(0, 3, 1, 2), xt.grad.numpy(), 1e-2, 1e-2 * float(xt.grad.abs().max()), 'deconv dgrad')
    # pick
    HW, K = 2 * H * 2 * W, O
    idx = torch.from_numpy(rs.randint(0, K, N).astype(np.float32)).to(dev())
    pk = torch.empty((N, HW), dtype=torch.float16, device=dev())
    hip.call('sn_pick_fwd', out, idx, pk, N, HW, K, hip.stream())
    ref = out.reshape(N, HW, K)
    assert all(torch.equal(pk[n], ref[n, :, int(idx[n])]) for n in range(N))
    g1 = torch.from_numpy(rs.standard_normal((N, HW)).astype(np.float32)).to(dev()).half()
    dxp = torch.full((N, HW, K), 7.0, dtype=torch.float16, device=dev())
    hip.call('sn_pick_bwd', g1, idx, dxp, N, HW, K, 0, hip.stream())
    want = torch.zeros((N, HW, K), dtype=torch.float16, device=dev())
    for n in range(N):
        want[n, :, int(idx[n])] = g1[n]
    assert torch.equal(dxp, want)
    idx2 = torch.from_numpy(((idx.cpu().numpy() + rs.randint(0, 2, N)) % K).astype(np.float32)).to(dev())
    hip.call('sn_pick_bwd', g1, idx2, dxp, N, HW, K, 1, hip.stream())
    for n in range(N):
        want[n, :, int(idx2[n])] = (want[n, :, int(idx2[n])].float() + g1[n].float()).half()
    assert torch.equal(dxp, want)


def test_multi_proposal_target_mask_vs_oracle():
    hip = _hip()
    rs = np.random.RandomState(6)
    B, A, Fh, Fw, stride, pre, post, G, nm = 3, 21, 16, 16, 16, 600, 60, 100, 5
    cls_prob = rs.uniform(0, 1, (B, 2, A * Fh, Fw)).astype(np.float32)
    bbox_pred = (rs.standard_normal((B, 4 * A, Fh, Fw)) * 0.3).astype(np.float32)
    im_info = np.array([[Fh * 16, Fw * 16, 1.0]] * B, np.float32)
    gt = -np.ones((B, G, 5), np.float32)
    for b, n in enumerate((14, 1, 0)):                 # many / few / no foreground RoIs
        c = rs.uniform(30, Fh * 16 - 30, (n, 2))
        wh = rs.uniform(30, 150, (n, 2))
        gt[b, :n, :4] = np.concatenate((c - wh / 2, c + wh / 2), 1)
        gt[b, :n, 4] = rs.randint(1, 81, n)
    vr = np.array([[0, 256]] * B, np.float32)
    from sniper_amd.data.anchors import generate_anchors
    base = generate_anchors(stride, [0.5, 1, 2], np.array((2, 4, 7, 10, 13, 16, 24), np.float32)).astype(np.float32)
    td = lambda z: torch.from_numpy(np.ascontiguousarray(z)).to(dev())
    ws = torch.empty(hip.query('sn_proposal_workspace_bytes', B, A, Fh, Fw, pre, post), dtype=torch.uint8, device=dev())
    match = torch.empty((B * post,), device=dev())
    rois, label = torch.empty((B * post, 5), device=dev()), torch.empty((B * post,), device=dev())
    tgt, wgt = torch.empty((B * post, 4), device=dev()), torch.empty((B * post, 4), device=dev())
    mrois, mids = torch.full((B * nm, 5), 7.0, device=dev()), torch.full((B * nm,), 7.0, device=dev())
    stds = np.array([0.1, 0.1, 0.2, 0.2], np.float32)
    hip.call('sn_multi_proposal_target_mask', td(cls_prob), td(bbox_pred), td(im_info), td(gt), td(vr), td(base), B, A, Fh, Fw, stride, G,
             pre, post, 0.7, 0.0, 0.5, stds.ctypes.data, ws, match, nm, rois, label, tgt, wgt, mrois, mids, hip.stream())
    torch.cuda.synchronize()
    r, lab = rois.cpu().numpy(), label.cpu().numpy()
    wl, wt, ww = onn.proposal_targets(r, gt, vr, post)
    assert np.array_equal(lab, wl)
    wm = onn.proposal_target_matches(r, gt, vr, post)
    assert np.array_equal(match.cpu().numpy(), wm)
    wr, wi = onn.mask_rois_select(r, lab, wm, post, nm)
    assert np.array_equal(mrois.cpu().numpy(), wr) and np.array_equal(mids.cpu().numpy(), wi)
    nfg = [(lab[b * post:(b + 1) * post] > 0).sum() for b in range(B)]
    assert nfg[0] > nm and nfg[1] < nfg[0] and nfg[2] == 0, nfg      # truncation, padding, all-padding


def test_mask_training_step_through_the_iterator():
    """configs/faster/sniper_res101_e2e_mask.yml pieces end to end: roidb with polygons -> chips -> gt_masks labels ->
    mask network -> losses -> SGD.  Outputs finite, the mask head receives gradient, targets look like masks."""
    from sniper_amd import config as cfgmod
    from sniper_amd.train import Trainer
    cfg = cfgmod.res101_e2e_mask(batch_images=2)
    tr = Trainer(batch_images=2, n_images=4, seed=0, cfg=cfg)
    assert [k for k, _ in tr.iter.provide_label][-1] == 'gt_masks' and dict(tr.iter.provide_label)['gt_masks'] == (2, 100, 500)
    before = tr.mod.exe.params['mask_out_weight'].master.clone()
    outs = tr.step()
    torch.cuda.synchronize()
    assert len(outs) == 7 and tuple(outs[5].shape) == (100, 2, 28, 28) and tuple(outs[6].shape) == (100, 28, 28)
    assert all(np.isfinite(o.asnumpy()).all() for o in outs)
    t = outs[6].asnumpy()
    assert set(np.unique(t).tolist()) <= {-1.0, 0.0, 1.0}
    p = outs[5].asnumpy()
    assert np.allclose(p.sum(1), 1.0, atol=1e-3)
    if (t >= 0).any():
        assert float((tr.mod.exe.params['mask_out_weight'].master - before).abs().sum()) > 0


def test_r101_mask_network_parity_vs_cpu_reference_ops():
    """The mask network at 2 chips, teacher-forced against oracle/graph_cpu.py like C1 / C2 / C4 (tests/test_gpu_engine.py)."""
    from test_gpu_engine import _forced_parity, _init_params, _train_inputs
    from sniper_amd import config as cfgmod
    from sniper_amd.data import mask_utils
    from sniper_amd.engine.executor import Executor
    from sniper_amd.symbols.faster import resnet_mx_101_e2e_mask as mk
    from sniper_amd.synthetic import make_polygons
    from sniper_amd.train import fixed_param_names
    B, A, F = 2, 21, 32
    cfg = cfgmod.res101_e2e_mask(batch_images=B)
    sym = mk.resnet_mx_101_e2e_mask(momentum=0.995).get_symbol_rcnn(cfg)
    shapes = dict(data=(B, 3, 512, 512), valid_ranges=(B, 2), im_info=(B, 3), label=(B, A * F * F),
                  bbox_target=(B, 4 * A, F, F), bbox_weight=(B, 4 * A, F, F), gt_boxes=(B, 100, 5), gt_masks=(B, 100, 500))
    os.environ['SNIPER_HIP_GRAPHS'] = '0'
    try:
        ex = Executor(sym, shapes, True, fixed_param_names(cfg, sym))
    finally:
        os.environ.pop('SNIPER_HIP_GRAPHS', None)
    rs = np.random.RandomState(15)
    P, AUX = _init_params(sym, shapes, rs, bn_gamma=(0.5, 1.0), bn_beta=(-0.2, 0.4))
    for k in P:
        if k.startswith('mask_') and k.endswith('_weight') and 'offset' not in k:
            P[k] = f16r(rs.standard_normal(P[k].shape) * np.sqrt(2.0 / np.prod(P[k].shape[1:])))
    P['bn_data_gamma'][:] = 1.0
    AUX['bn_data_moving_mean'][:] = 0.0
    AUX['bn_data_moving_var'][:] = 1.0 - 2e-5
    P['bn_data_beta'][:] = 0.0
    inp = _train_inputs(rs, B, A, F)
    enc = -np.ones((B, 100, 500), np.float32)
    for b in range(B):
        n = int((inp['gt_boxes'][b, :, 4] >= 0).sum())
        polys = [make_polygons(rs, inp['gt_boxes'][b, g, :4]) for g in range(n)]
        enc[b] = mask_utils.poly_encoder(mask_utils.crop_polys(polys, [0, 0], 1.0), inp['gt_boxes'][b, :n, 4] - 1)
    inp['gt_masks'] = enc
    checked, ov = _forced_parity(sym, ex, P, AUX, inp, tol_fwd=2e-3, tol_grad=2e-2)
    assert checked >= 260
    mids = ov[('multi_proposal_target_mask', 5)]
    assert (mids >= 0).sum() >= 1, 'the synthetic GT must give some mask RoIs'
