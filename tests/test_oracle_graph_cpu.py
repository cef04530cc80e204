"""CPU: oracle/graph_cpu.py (the whole-graph fp32 evaluator used for the end-to-end parity runs) against a
hand-written torch restatement of a small SNIPER-shaped graph, and on the operator gradient rules MXNet defines."""
import numpy as np
import torch

import sniper_amd.mx as mx
from oracle import graph_cpu
from test_gpu_engine import _mini_graph, _torch_reference


def test_graph_cpu_matches_handwritten_reference():
    A, B, S = 3, 2, 64
    sym = _mini_graph(mx, A)
    F = S // 8
    shapes = dict(data=(B, 3, S, S), label=(B, A * F * F), bbox_target=(B, 4 * A, F, F), bbox_weight=(B, 4 * A, F, F))
    rs = np.random.RandomState(0)
    args, _, auxs = sym.infer_shape(**shapes)
    P, AUX = {}, {}
    for name, shp in zip(sym.list_arguments(), args):
        if name in shapes:
            continue
        if name.endswith('_gamma'):
            P[name] = rs.uniform(0.5, 1.5, shp).astype(np.float32)
        elif name.endswith('_beta') or name.endswith('_bias'):
            P[name] = (rs.standard_normal(shp) * 0.1).astype(np.float32)
        else:
            P[name] = (rs.standard_normal(shp) * np.sqrt(2.0 / np.prod(shp[1:]))).astype(np.float32)
    for name, shp in zip(sym.list_auxiliary_states(), auxs):
        AUX[name] = rs.uniform(0.5, 1.5, shp).astype(np.float32)
    P['bn_data_gamma'][:] = 1.0
    inp = dict(data=(rs.standard_normal((B, 3, S, S)) * 2).astype(np.float32),
               label=rs.choice([-1, 0, 1], size=(B, A * F * F), p=[0.5, 0.3, 0.2]).astype(np.float32),
               bbox_target=rs.standard_normal((B, 4 * A, F, F)).astype(np.float32),
               bbox_weight=(rs.uniform(size=(B, 4 * A, F, F)) < 0.2).astype(np.float32))
    # the hand-written reference rounds weights to fp16 first (what the device multiplies); feed the same values
    P16 = {k: (v.astype(np.float16).astype(np.float32) if v.ndim > 1 else v) for k, v in P.items()}
    want_prob, want_l1, want_g = _torch_reference(P, AUX, inp, A)
    outs, grads = graph_cpu.run(sym, P16, AUX, inp)
    assert np.allclose(outs[0], want_prob, rtol=1e-4, atol=1e-5)
    assert np.allclose(outs[1], want_l1, rtol=1e-4, atol=1e-5)
    n = 0
    for k, g in want_g.items():
        if g is None or k.startswith('bn_data') or k.startswith('bn0') or k.startswith('conv0'):
            continue      # frozen in the executor test; autograd still differentiates them here
        assert np.allclose(grads[k], g, rtol=2e-3, atol=2e-4 * np.abs(g).max()), k
        n += 1
    assert n >= 20


def test_graph_cpu_loss_gradient_rules():
    """SoftmaxOutput ignores the incoming gradient and normalises by the valid count; clip passes the gradient on the
    closed interval; MakeLoss injects grad_scale."""
    x = mx.sym.Variable('x')
    lab = mx.sym.Variable('lab')
    w = mx.sym.Variable('w_weight')
    y = mx.sym.clip(x * w, 0, 6, name='c')
    prob = mx.sym.SoftmaxOutput(data=y, label=lab, use_ignore=True, ignore_label=-1, normalization='valid', grad_scale=2.0,
                                name='p')
    loss = mx.sym.MakeLoss(data=y, grad_scale=0.5, name='l')
    sym = mx.sym.Group([prob, loss])
    xv = np.array([[-1.0, 0.0, 3.0], [6.0, 7.0, 2.0]], np.float32)
    wv = np.ones((2, 3), np.float32)
    lv = np.array([2, -1], np.float32)
    outs, grads = graph_cpu.run(sym, {'w_weight': wv}, {}, {'x': xv, 'lab': lv})
    yv = np.clip(xv, 0, 6)
    p = np.exp(yv) / np.exp(yv).sum(1, keepdims=True)
    assert np.allclose(outs[0], p, atol=1e-6)
    g_y = np.zeros((2, 3), np.float32)
    g_y[0] = (p[0] - np.eye(3)[2]) * 2.0 / 1.0       # one valid row -> normaliser 1; row 1 is ignored
    g_y += 0.5                                        # MakeLoss
    mask = (xv >= 0) & (xv <= 6)                      # clip: closed interval
    assert np.allclose(grads['w_weight'], g_y * mask * xv, atol=1e-6)


def test_position_sensitive_pooling_is_the_diagonal_of_group1_pooling():
    """oracle/nn.py dpsroi_pool(group_size=G): bin (ph,pw) of output channel d must equal the group_size-1 pooling of map
    channel (d*G+ph)*G+pw at the same bin -- forward, data gradient and offset gradient (BASELINE config C4)."""
    from oracle import nn as onn
    rs = np.random.RandomState(3)
    B, D, G, H, W, R, S = 2, 3, 3, 9, 11, 5, 2
    P = G
    C = D * G * G
    data = rs.standard_normal((B, C, H, W))
    rois = np.zeros((R, 5), np.float32)
    rois[:, 0] = rs.randint(0, B, R)
    x1, y1 = rs.uniform(-10, 120, R), rs.uniform(-10, 90, R)
    rois[:, 1], rois[:, 2], rois[:, 3], rois[:, 4] = x1, y1, x1 + rs.uniform(8, 90, R), y1 + rs.uniform(8, 70, R)
    trans = rs.uniform(-1, 1, (R, 2, P, P)).astype(np.float32)
    diag = np.zeros((C, P, P), bool)
    for d in range(D):
        for ph in range(P):
            for pw in range(P):
                diag[(d * G + ph) * G + pw, ph, pw] = True
    for tr in (None, trans):
        out = onn.dpsroi_pool(data, rois, tr, P, S, 1.0 / 8, 0.1, group_size=G)
        full = onn.dpsroi_pool(data, rois, tr, P, S, 1.0 / 8, 0.1)
        assert out.shape == (R, D, P, P)
        assert np.allclose(out.reshape(R, -1), full[:, diag].reshape(R, D, P, P).reshape(R, -1), atol=1e-12)
        dout = rs.standard_normal(out.shape)
        dfull = np.zeros(full.shape)
        dfull[:, diag] = dout.reshape(R, -1)
        dd, dt = onn.dpsroi_pool_backward(dout, data, rois, tr, P, S, 1.0 / 8, 0.1, group_size=G)
        dd1, dt1 = onn.dpsroi_pool_backward(dfull, data, rois, tr, P, S, 1.0 / 8, 0.1)
        assert np.allclose(dd, dd1, atol=1e-12)
        assert (dt is None and dt1 is None) or np.allclose(dt, dt1, atol=1e-12)
    assert np.abs(out).max() > 0 and np.abs(dd).max() > 0 and np.abs(dt).max() > 0


def test_dpsroi_sparse_operator_form_equals_the_loop_definition():
    """oracle/nn.py dpsroi_pool_fast / dpsroi_pool_backward_fast (three sparse sampling matrices; used at BASELINE sizes, where
    the loop definition takes minutes) against dpsroi_pool / dpsroi_pool_backward: group_size 1 and 3, with and without learned
    offsets, RoIs hanging over every border, degenerate (sub-pixel) RoIs."""
    from oracle import nn as onn
    rs = np.random.RandomState(0)
    for G, C, P in ((1, 16, 7), (3, 36, 6), (7, 49 * 3, 7)):
        B, H, W, S, R = 2, 12, 14, 4, 15
        data = rs.standard_normal((B, C, H, W)).astype(np.float32)
        c, wh = rs.uniform(0, 110, (R, 2)), rs.uniform(4, 90, (R, 2))
        rois = np.concatenate((rs.randint(0, B, (R, 1)), c - wh / 2, c + wh / 2), 1).astype(np.float32)
        rois[0, 1:] = [-30, -20, 5, 8]
        rois[1, 1:] = [100, 90, 140, 130]
        rois[2, 1:] = [33.2, 47.6, 33.4, 47.9]
        D = C // (G * G)
        for tr in (None, (rs.standard_normal((R, 2, P, P)) * 0.6).astype(np.float32)):
            a = onn.dpsroi_pool(data, rois, tr, P, S, 1 / 8., 0.1, G)
            b = onn.dpsroi_pool_fast(data, rois, tr, P, S, 1 / 8., 0.1, G)
            assert np.abs(a - b).max() <= 1e-6 * max(1.0, np.abs(a).max())
            dout = rs.standard_normal((R, D, P, P))
            d1, t1 = onn.dpsroi_pool_backward(dout, data, rois, tr, P, S, 1 / 8., 0.1, G)
            d2, t2 = onn.dpsroi_pool_backward_fast(dout, data, rois, tr, P, S, 1 / 8., 0.1, G)
            assert np.abs(d1 - d2).max() <= 1e-6 * max(1.0, np.abs(d1).max())
            if tr is not None:
                assert np.abs(t1 - t2).max() <= 1e-5 * max(1.0, np.abs(t1).max())


def test_deformable_sampling_array_form_equals_the_loop_statement():
    """oracle/nn.py: deform_im2col / deform_col2im walk (image, group) with array operations; the per-sample loops they replace stay
    as deform_*_loops -- the statement (one sample at a time, the branches of _deform_sample spelled out).  Equal on dilated,
    strided and 1 x 1 kernels, offsets that leave the map, positions that land exactly on the border."""
    import numpy as np
    from oracle import nn
    rs = np.random.RandomState(0)
    for (N, C, H, W, K, stride, pad, dil, DG) in ((2, 8, 9, 11, 3, 1, 2, 2, 4), (1, 4, 7, 7, 3, 2, 1, 1, 1), (2, 12, 6, 10, 1, 1, 0, 1, 2)):
        T = K * K
        Ho = (H + 2 * pad - dil * (K - 1) - 1) // stride + 1
        Wo = (W + 2 * pad - dil * (K - 1) - 1) // stride + 1
        data = rs.standard_normal((N, C, H, W))
        off = (rs.standard_normal((N, 2 * T * DG, Ho, Wo)) * 2.5).astype(np.float32)
        off[0, 0, 0, 0] = np.float32(H)                       # far outside
        off[0, 1, 0, 0] = np.float32(W - 1 + pad)             # exactly on the last column for tap 0
        a, b = nn.deform_im2col(data, off, K, K, stride, pad, dil, DG), nn.deform_im2col_loops(data, off, K, K, stride, pad, dil, DG)
        assert np.array_equal(a, b)
        dcol = rs.standard_normal(a.shape)
        a1, a2 = nn.deform_col2im(dcol, data, off, K, K, stride, pad, dil, DG)
        b1, b2 = nn.deform_col2im_loops(dcol, data, off, K, K, stride, pad, dil, DG)
        assert np.abs(a1 - b1).max() < 1e-12 and np.abs(a2 - b2).max() < 1e-12
