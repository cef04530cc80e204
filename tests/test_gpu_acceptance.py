"""-m gpu: the reference's own main_train.py and main_test.py executed over sniper_amd (tests/acceptance_main_{train,test}.py) -- SURVEY section 8(b):
"run them over our mxnet shim -- this *is* the acceptance test".  The reference's Python travels to the GPU box as the
lib2to3 artefact oracle/_ref/py3 (oracle/build.py::build_reference_py3; git-ignored like the compiled reference modules)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_main_train_runs_unchanged(tmp_path):
    if not os.path.isfile(os.path.join(ROOT, 'oracle', '_ref', 'py3', 'main_train.py')):
        pytest.skip('oracle/_ref/py3 not built (python -m oracle.build where the reference checkout exists)')
    out = str(tmp_path / 'acceptance.json')
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'acceptance_main_train.py'), out], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-6000:]
    res = json.load(open(out))
    # main_train.py:36-146 ran to the end of its epoch on the reference's iterator and symbol
    assert res['iterator'] == 'iterators.PrefetchingIter.PrefetchingIter' and res['symbol'] == 'symbols.faster.resnet_mx_101_e2e'
    assert res['batches'] >= 3, res
    m = res['metrics']
    for name in ('RPNAcc', 'RPNLogLoss', 'RPNL1Loss', 'RCNNAcc', 'RCNNLogLoss', 'RCNNL1LossCRCNN'):
        assert name in m and np.isfinite(m[name]), m
    assert 0.0 <= m['RPNAcc'] <= 1.0 and m['RPNLogLoss'] > 0
    # trainable parameters moved, the frozen stage did not (FIXED_PARAMS, yml:22-25)
    d = res['param_delta']
    assert d['rpn_conv_3x3_weight'] > 0 and d['stage3_unit1_conv1_weight'] > 0 and d['fc_new_1_weight'] > 0
    assert d['stage1_unit1_conv1_weight'] == 0.0
    # mx.callback.module_checkpoint + the symbol file's checkpoint_callback wrote an MXNet-format checkpoint with the de-normalised
    # box regression weights (resnet_mx_101_e2e.py:8-19)
    assert any(f.endswith('.params') for f in res['checkpoint_files']), res['checkpoint_files']
    assert res['checkpoint_keys'] > 500 and res['checkpoint_has_test_weights']
    # the reference anchor_worker's labels == the GPU labelling of the same chips
    assert len(res['anchor_labels']) >= 2
    for c in res['anchor_labels']:
        assert c['label_equal'] and c['weight_equal'] and c['gt_equal'] and c['target_maxdiff'] <= 1e-6, c


def test_reference_main_test_runs_unchanged(tmp_path):
    """main_test.py:32-61 -> lib/inference.py imdb_detection_wrapper / detect_scale_worker / Tester over the shim: three test
    scales, per-scale pickles, valid-range aggregation, soft-NMS through the cpu_nms mirror (bit-equal to the reference's compiled
    cpu_nms.pyx on the recorded calls)."""
    if not os.path.isfile(os.path.join(ROOT, 'oracle', '_ref', 'py3', 'main_test.py')):
        pytest.skip('oracle/_ref/py3 not built (python -m oracle.build where the reference checkout exists)')
    out = str(tmp_path / 'acceptance_test.json')
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'acceptance_main_test.py'), out], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-6000:]
    res = json.load(open(out))
    assert res['evaluated'] and res['classes'] == 81 and res['images'] == res['n_images'] and res['outputs_finite']
    # every test scale of the yml ran (three different input extents reached Module.forward) and cached its detections
    assert len(res['forward_shapes']) >= 3, res['forward_shapes']
    for d in ('dets_scale_1400x2000', 'dets_scale_800x1280', 'dets_scale_480x512'):
        assert d in res['scales'] and res['scales'][d]['dets'] > 0, res['scales']
        assert res['scales'][d]['min_score'] > 1e-3                    # Tester.get_detections' score threshold (inference.py:290)
    assert 'dets_final/detections.pkl' in res['result_files'] and res['final_pickle_equal']
    assert res['shape_ok'] and res['finite'] and res['in_bounds']
    # MAX_PER_IMAGE = 200 (yml:177) keeps scores >= the 200th largest: ties may add a few
    assert all(0 < n <= 220 for n in res['dets_per_image']), res['dets_per_image']
    assert res['nms_replayed'] >= 10 and res['nms_bit_equal'] == res['nms_replayed'], res


def test_reference_main_test_extracts_proposals(tmp_path):
    """TEST.EXTRACT_PROPOSALS: main_test.py:58-59 -> imdb_proposal_extraction_wrapper / proposal_scale_worker /
    Tester.extract_proposals (lib/inference.py:556-609,531-553,372-408) with the reference's MNIteratorTest and
    get_symbol_rpn(is_train=False); writes the `<DATASET>_<image set>_rpn.pkl` negative-chip proposal file (list of (n, 5) float32
    [x1, y1, x2, y2, score], the format lib/dataset/imdb.py:81-118 reads back)."""
    if not os.path.isfile(os.path.join(ROOT, 'oracle', '_ref', 'py3', 'main_test.py')):
        pytest.skip('oracle/_ref/py3 not built (python -m oracle.build where the reference checkout exists)')
    out = str(tmp_path / 'acceptance_props.json')
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'acceptance_main_test.py'), out, '--proposals'], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-6000:]
    res = json.load(open(out))
    assert res['proposal_files'] == ['COCO_synthetic_val_rpn.pkl'] and res['images'] == res['n_images']
    assert res['shapes'] == [[900, 5]] * res['n_images'] and res['dtype'] == 'float32'        # 3 scales x N_PROPOSAL_PER_SCALE (yml:250)
    assert res['in_bounds'] and res['scores_sorted']
    assert res['scale_files'] == ['props_scale_1400x2000', 'props_scale_480x512', 'props_scale_800x1280']
    assert len(res['forward_shapes']) >= 3


def _ohem_numpy(cls_score, bbox_pred, labels, bbox_targets, bbox_weights, k):
    """numpy restatement of BoxAnnotatorOHEMOperator.forward (lib/operator_py/box_annotator_ohem.py:27-63): per image keep the k
    RoIs with the largest (softmax cross-entropy + smooth-L1) loss among the labelled ones, ignore the rest."""
    lab_out, w_out = labels.copy(), bbox_weights.copy()
    for i in range(labels.shape[0]):
        z = cls_score[i] - cls_score[i].max(1, keepdims=True)
        p = np.exp(z) / np.exp(z).sum(1, keepdims=True) + 1e-14
        nv = labels[i] < 0
        li = np.where(nv, 0, labels[i]).astype(int)
        lc = -np.log(p[np.arange(p.shape[0]), li])
        lc[nv] = 0
        d = bbox_pred[i] - bbox_targets[i]
        sl1 = np.where(np.abs(d) < 1, 0.5 * d * d, np.abs(d) - 0.5)
        lb = (bbox_weights[i] * sl1).sum(1)
        lb[nv] = 0
        order = np.argsort(lc + lb)
        drop = order[::-1][k:]
        lab_out[i][drop] = -1
        w_out[i][drop] = 0
    return lab_out, w_out


def test_reference_operator_plugin_executes_in_a_graph():
    """lib/operator_py/box_annotator_ohem.py, the reference's operator-plugin exemplar, registered through mx.operator.register
    and placed in a graph with mx.sym.Custom the way symbols/faster/resnext_mx_101.py:312-331 wires it: its forward runs on the
    tensors the device graph produced, its backward hands zeros to its inputs, and the losses that consume its outputs train
    the layers below with exactly those (constant) selections."""
    py3 = os.path.join(ROOT, 'oracle', '_ref', 'py3')
    if not os.path.isfile(os.path.join(py3, 'lib', 'operator_py', 'box_annotator_ohem.py')):
        pytest.skip('oracle/_ref/py3 not built')
    import importlib
    import torch
    import sniper_amd.mx as mx
    from sniper_amd.engine.executor import Executor
    mx.alias_as('mxnet')
    sys.path.insert(0, os.path.join(py3, 'lib'))
    try:
        importlib.import_module('operator_py.box_annotator_ohem')      # @mx.operator.register('BoxAnnotatorOHEM')
    finally:
        sys.path.remove(os.path.join(py3, 'lib'))
    B, R, D, C, K = 2, 24, 64, 5, 7
    data = mx.sym.Variable('data')
    label, bt, bw = mx.sym.Variable('label'), mx.sym.Variable('bbox_target'), mx.sym.Variable('bbox_weight')
    cls_score = mx.sym.FullyConnected(name='cls_score', data=data, num_hidden=C)
    bbox_pred = mx.sym.FullyConnected(name='bbox_pred', data=data, num_hidden=4)
    cs3 = mx.sym.Reshape(data=cls_score, shape=(-1, R, C), name='cls_score_3d')
    bp3 = mx.sym.Reshape(data=bbox_pred, shape=(-1, R, 4), name='bbox_pred_3d')
    labels_ohem, bbox_weights_ohem = mx.sym.Custom(op_type='BoxAnnotatorOHEM', num_classes=C, num_reg_classes=1, roi_per_img=K,
                                                   cls_score=cs3, bbox_pred=bp3, labels=label, bbox_targets=bt, bbox_weights=bw)
    lab_flat = mx.sym.Reshape(data=labels_ohem, shape=(-1,), name='label_reshape')
    w_flat = mx.sym.Reshape(data=bbox_weights_ohem, shape=(-1, 4), name='bbox_weight_reshape')
    t_flat = mx.sym.Reshape(data=bt, shape=(-1, 4), name='bbox_target_reshape')
    cls_prob = mx.sym.SoftmaxOutput(name='cls_prob', data=cls_score, label=lab_flat, normalization='valid', use_ignore=True,
                                    ignore_label=-1, grad_scale=1.0)
    bbox_loss_ = w_flat * mx.sym.smooth_l1(name='bbox_loss_', scalar=1.0, data=(bbox_pred - t_flat))
    bbox_loss = mx.sym.MakeLoss(name='bbox_loss', data=bbox_loss_, grad_scale=1.0 / (K * B))
    sym = mx.sym.Group([cls_prob, bbox_loss, mx.sym.BlockGrad(lab_flat)])
    shapes = dict(data=(B * R, D), label=(B, R), bbox_target=(B, R, 4), bbox_weight=(B, R, 4))
    ex = Executor(sym, shapes, True, [])
    rs = np.random.RandomState(0)
    P = {'cls_score_weight': (rs.standard_normal((C, D)) * 0.3).astype(np.float32), 'cls_score_bias': np.zeros(C, np.float32),
         'bbox_pred_weight': (rs.standard_normal((4, D)) * 0.1).astype(np.float32), 'bbox_pred_bias': np.zeros(4, np.float32)}
    ex.set_params(P, {})
    x = rs.standard_normal((B * R, D)).astype(np.float32)
    lab = rs.choice([-1, 0, 1, 2, 3, 4], size=(B, R), p=[0.2, 0.4, 0.1, 0.1, 0.1, 0.1]).astype(np.float32)
    tgt = rs.standard_normal((B, R, 4)).astype(np.float32)
    wgt = np.repeat((lab > 0).astype(np.float32)[:, :, None], 4, 2)
    outs = ex.forward(dict(data=x, label=lab, bbox_target=tgt, bbox_weight=wgt), is_train=True)
    ex.backward()
    torch.cuda.synchronize()
    # forward of the plugin == the numpy restatement, on the scores the device produced (fp16 operands -> recompute from them)
    from gpu_util import f16r
    xf, Wc, Wb = f16r(x), f16r(P['cls_score_weight']), f16r(P['bbox_pred_weight'])
    score, pred = xf @ Wc.T, xf @ Wb.T
    want_lab, want_w = _ohem_numpy(score.reshape(B, R, C), pred.reshape(B, R, 4), lab, tgt, wgt, K)
    got_lab = outs[2].cpu().numpy().reshape(B, R)
    # the selection is an arg-sort of losses: equal unless two losses tie within fp16 rounding of the device scores
    assert (got_lab == want_lab).mean() >= 0.95, (got_lab, want_lab)
    for i in range(B):
        assert int((got_lab[i] >= 0).sum()) == min(K, int((lab[i] >= 0).sum()))
    # backward: the plugin contributes zeros, so the parameter gradients are those of the two losses with the plugin's
    # outputs held constant (torch autograd on the same fp16-rounded operands, selections taken from the device run)
    sel_lab = torch.from_numpy(got_lab.reshape(-1)).long()
    sel_w = torch.from_numpy(np.where((got_lab >= 0)[:, :, None], wgt, 0).astype(np.float32).reshape(-1, 4))
    xt = torch.from_numpy(xf)
    wc = torch.from_numpy(Wc).requires_grad_(True)
    wb = torch.from_numpy(Wb).requires_grad_(True)
    logp = torch.log_softmax(xt @ wc.t(), 1)
    valid = sel_lab >= 0
    ce = -(logp[torch.arange(B * R), sel_lab.clamp(min=0)] * valid).sum() / max(1, int(valid.sum()))
    d = xt @ wb.t() - torch.from_numpy(tgt.reshape(-1, 4))
    sl1 = (torch.where(d.abs() < 1, 0.5 * d * d, d.abs() - 0.5) * sel_w).sum() / (K * B)
    (ce + sl1).backward()
    from gpu_util import assert_close
    for name, want in (('cls_score_weight', wc.grad.numpy()), ('bbox_pred_weight', wb.grad.numpy())):
        p = ex.params[name]
        got = p.to_reference(p.grad.detach().cpu().numpy())
        assert_close(got, want, 1e-2, 1e-2 * np.abs(want).max(), 'grad ' + name)


def test_reference_epoch_chip_database_through_the_routed_pool():
    """The reference's unchanged MNIteratorE2E.reset (lib/iterators/MNIteratorE2E.py:40-103) with its own chip_worker over the
    extension mirrors: the drop-in pool (sniper_amd/ext/pool.py) runs each `pool.map(chip_worker.chip_extractor / box_assigner,
    part)` as ONE ragged GPU launch instead of one launch + read-back per image on a pool thread.  Same numpy seed -> the routed
    and the per-item build produce the SAME chip database (chips, scales, box assignment)."""
    if not os.path.isfile(os.path.join(ROOT, 'oracle', '_ref', 'py3', 'main_train.py')):
        pytest.skip('oracle/_ref/py3 not built (python -m oracle.build where the reference checkout exists)')
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'chipdb_bench.py'), '300', '120'], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert res['pool'] == 'sniper_amd.ext.pool.Pool', res
    # (the yml splits the roidb into TRAIN.CHIPS_DB_PARTS = 20 parts: two maps per part)
    assert res['routed_maps'] == 40 and res['unrouted']['routed_maps'] == 0, res
    assert res['routed_equals_unrouted'] and res['chips'] > 300, res


def test_reference_get_batch_through_the_routed_per_batch_maps():
    """The reference's unchanged MNIteratorE2E._get_batch (lib/iterators/MNIteratorE2E.py:112-220): its two per-batch maps --
    `pool.map(anchor_worker.worker, ...)` and `thread_pool.map_async(im_worker.worker, ...)` -- run as the mirrors' batched GPU work
    and its own assembly lines (`mx.nd.zeros` + per-chip writes) then build the batch IN HBM (sniper_amd/ext/pool.py, the shim's
    unplaced arrays).  Against the same call with the routing off (the reference's numpy workers): ground-truth rows, valid ranges
    and im_info equal; foreground anchors, box weights and targets equal wherever no random draw is involved; pixels of the same
    crops (cv2-exact resize vs the harness's PIL stand-in: close, not equal)."""
    if not os.path.isfile(os.path.join(ROOT, 'oracle', '_ref', 'py3', 'main_train.py')):
        pytest.skip('oracle/_ref/py3 not built (python -m oracle.build where the reference checkout exists)')
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'routed_batch_check.py'), '24', '8'], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert res['pool'] == 'sniper_amd.ext.pool.Pool' and res['thread_pool'] == 'sniper_amd.ext.pool.Pool', res
    checked = 0
    for b in res['batches']:
        assert b['routed_maps'] == [1, 1], b                      # one anchor map, one image map per batch
        assert all(b['routed_on_device']) and not any(b['unrouted_on_device']), b
        assert b['gt_equal'] and b['valid_ranges_equal'] and b['im_info_equal'], b
        assert b['pixels_corr'] > 0.98 and b['pixels_mean_abs_diff'] < 6.0, b
        for c in b['chips']:
            assert c['bg'][0] == 256 - c['fg'][0] or c['fg'][0] + c['bg'][0] <= 256, c
            if c['fg_equal'] is not None:
                assert c['fg_equal'] and c['weights_equal'] and c['targets_maxdiff'] <= 1e-5, c
                checked += 1
    assert checked >= 6          # (chips with 128+ foreground candidates involve a draw and are not compared anchor by anchor)
    # SNIPER_NUMPY_RNG=1: numpy's own draws replayed (data_workers.py:327-338) -> under np.random.seed the routed batch equals the
    # unrouted reference call on EVERY chip, the sub-sampled ones included (north_star: bit-exact index sets on fixed seeds)
    drawn = 0
    for b in res['batches']:
        n = b['numpy_rng']
        assert all(n['on_device']), n
        assert n['labels_equal'] and n['weights_equal'] and n['gt_equal'] and n['targets_maxdiff'] <= 1e-5, n
        drawn += n['chips_with_a_draw']
    assert drawn >= 8 and any(b['numpy_rng']['differs_from_hashed_draws'] for b in res['batches']), res
