"""CPU, needs the reference checkout: the reference's OWN training utilities (lib/train_utils/{metric,utils,lr_scheduler}.py,
the pieces main_train.py:97-146 wires together) imported unchanged over sniper_amd.mx -- the claim "drops into main_train.py".
Outputs are checked against plain numpy restatements of what each metric computes."""
import importlib
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.ref


@pytest.fixture(scope='module')
def ref():
    import sniper_amd.mx as mx
    mx.alias_as('mxnet')
    for p in ('/root/reference', '/root/reference/lib'):
        if p not in sys.path:
            sys.path.insert(0, p)
    sys.dont_write_bytecode = True
    ns = type('ns', (), {})()
    ns.mx = mx
    ns.metric = importlib.import_module('train_utils.metric')
    ns.utils = importlib.import_module('train_utils.utils')
    ns.sched = importlib.import_module('train_utils.lr_scheduler')
    return ns


def _cfg(with_mask=False, auto_focus=False):
    from sniper_amd import config as cfgmod
    cfg = cfgmod.res101_e2e_mask(2) if with_mask else cfgmod.res101_e2e(2)
    cfg.TRAIN.ENABLE_OHEM = False
    cfg.TRAIN.END2END = True
    cfg.TRAIN.AUTO_FOCUS = auto_focus
    return cfg


def _fake_step(mx, rs, B=2, A=21, F=8, R=30, C=81, with_mask=False):
    """labels / preds shaped like the iterator's labels and the training graph's outputs"""
    nd = lambda a: mx.nd.array(np.asarray(a, np.float32))
    rpn_label = rs.choice([-1, 0, 1], size=(B, A * F * F), p=[0.8, 0.15, 0.05])
    labels = [nd(rpn_label), nd(rs.standard_normal((B, 4 * A, F, F))), nd(rs.uniform(size=(B, 4 * A, F, F)) < 0.1), nd(-np.ones((B, 100, 5)))]
    p = rs.uniform(0.01, 1, (B, 2, A * F, F))
    p /= p.sum(1, keepdims=True)
    rpn_loss = np.abs(rs.standard_normal((B, 4 * A, F, F))) * 0.01
    cls = rs.uniform(0.01, 1, (B, R, C))
    cls /= cls.sum(2, keepdims=True)
    bbox_loss = np.abs(rs.standard_normal((B, R, 4))) * 0.1
    rlabel = rs.choice([-1, 0, 3, 17], size=(B * R,), p=[0.1, 0.6, 0.15, 0.15])
    preds = [nd(p), nd(rpn_loss), nd(cls), nd(bbox_loss), nd(rlabel)]
    raw = dict(rpn_label=rpn_label, rpn_prob=p, rpn_loss=rpn_loss, cls=cls, bbox_loss=bbox_loss, rlabel=rlabel)
    if with_mask:
        mp = rs.uniform(0.01, 1, (B * 5, 2, 28, 28))
        mp /= mp.sum(1, keepdims=True)
        mt = rs.choice([-1, 0, 1], size=(B * 5, 28, 28), p=[0.3, 0.4, 0.3])
        preds += [nd(mp), nd(mt)]
        raw.update(mask_prob=mp, mask_t=mt)
    return labels, preds, raw


def test_reference_metrics_over_the_shim(ref):
    mx, m = ref.mx, ref.metric
    cfg = _cfg(with_mask=True)
    rs = np.random.RandomState(0)
    labels, preds, raw = _fake_step(mx, rs, with_mask=True)
    comp = mx.metric.CompositeEvalMetric()
    parts = [m.RPNAccMetric(), m.RPNLogLossMetric(), m.RPNL1LossMetric(), m.RCNNAccMetric(cfg), m.RCNNLogLossMetric(cfg),
             m.RCNNL1LossCRCNNMetric(cfg), m.MaskLogLossMetric(cfg)]
    for q in parts:
        comp.add(q)
    comp.update(labels, preds)
    got = dict(zip(*comp.get())) if isinstance(comp.get()[0], list) else dict([comp.get()])
    # numpy restatements
    lab = raw['rpn_label'].reshape(-1)
    pp = raw['rpn_prob'].reshape(2, 2, -1).transpose(0, 2, 1).reshape(-1, 2)
    keep = lab != -1
    assert np.isclose(got['RPNAcc'], (pp.argmax(1)[keep] == lab[keep]).mean())
    assert np.isclose(got['RPNLogLoss'], -np.log(pp[keep, lab[keep].astype(int)] + 1e-14).mean(), rtol=1e-5)
    rl = raw['rlabel']
    cp = raw['cls'].reshape(-1, raw['cls'].shape[-1])
    k2 = rl != -1
    assert np.isclose(got['RCNNAcc'], (cp.argmax(1)[k2] == rl[k2]).mean())
    assert np.isclose(got['RCNNLogLoss'], -np.log(cp[k2, rl[k2].astype(int)] + 1e-14).mean(), rtol=1e-5)
    mt = raw['mask_t'].reshape(-1)
    mp = raw['mask_prob'].reshape(raw['mask_prob'].shape[0], 2, -1).transpose(0, 2, 1).reshape(-1, 2)
    k3 = mt != -1
    assert np.isclose(got['MaskLogLoss'], -np.log(mp[k3, mt[k3].astype(int)] + 1e-14).mean(), rtol=1e-5)
    assert got['RPNL1Loss'] > 0 and got['RCNNL1LossCRCNN'] > 0
    comp.reset()
    assert all(np.isnan(v) or v == 0 for v in (dict(zip(*comp.get())) if isinstance(comp.get()[0], list) else {}).values())


def test_reference_optimizer_params_and_scheduler(ref):
    cfg = _cfg()
    iters, bs = 100000, 16                     # chips per epoch, global batch: the lr step (epoch 5.33) lies after the warm-up
    tr = cfg.TRAIN
    warm0 = tr.warmup_lr                       # get_optim_params rescales cfg.TRAIN.warmup_lr in place (utils.py:22-23)
    op = ref.utils.get_optim_params(cfg, iters, bs)
    assert op['multi_precision'] is True and np.isclose(op['learning_rate'], tr.lr / tr.scale) and np.isclose(op['wd'], tr.wd * tr.scale)
    assert op['momentum'] == tr.momentum and op['rescale_grad'] == 1.0 and op['clip_gradient'] is None
    assert np.isclose(tr.warmup_lr, warm0 / tr.scale)
    sch = op['lr_scheduler']
    assert isinstance(sch, ref.sched.WarmupMultiBatchScheduler) and isinstance(sch, ref.mx.lr_scheduler.LRScheduler)
    # the optimizer hands its learning rate to the scheduler (mx.optimizer: lr_scheduler.base_lr = learning_rate), which is
    # what Module.init_optimizer of the shim does too
    import sniper_amd.mx as mx
    from sniper_amd.symbols.faster import resnet_mx_101_e2e as ours
    mod = mx.mod.Module(ours.resnet_mx_101_e2e().get_symbol_rpn(cfg), data_names=['data'], label_names=['label', 'bbox_target', 'bbox_weight'])
    mod.init_optimizer(optimizer='sgd', optimizer_params=op)
    lr0, lr1, ws = warm0 / tr.scale, tr.lr / tr.scale, tr.warmup_step
    assert np.isclose(sch.base_lr, lr1)
    # linear warm-up from warmup_lr to lr over warmup_step updates, then the lr_step decay (lr_scheduler.py:43-66)
    assert np.isclose(sch(1), lr0 + (lr1 - lr0) * 1 / ws)
    assert np.isclose(sch(ws // 2), lr0 + (lr1 - lr0) * (ws // 2) / ws)
    assert np.isclose(sch(ws + 1), lr1)
    step = int(float(tr.lr_step) * iters / bs)
    assert step > ws and np.isclose(sch(step), lr1) and np.isclose(sch(step + 1), lr1 * tr.lr_factor)


def test_reference_fixed_params_and_checkpoint_loader(ref, tmp_path):
    from sniper_amd.symbols.faster import resnet_mx_101_e2e as ours
    from sniper_amd.train import fixed_param_names
    mx = ref.mx
    cfg = _cfg()
    sym = ours.resnet_mx_101_e2e().get_symbol_rcnn(cfg)
    theirs = ref.utils.get_fixed_param_names(cfg.network.FIXED_PARAMS, sym)
    assert sorted(theirs) == sorted(fixed_param_names(cfg, sym)) and 'conv0_weight' in theirs and 'stage1_unit1_bn1_gamma' in theirs
    # a checkpoint written through the shim (MXNet's binary container) read back by the reference's own load_param
    rs = np.random.RandomState(1)
    arg = {'conv0_weight': mx.nd.array(rs.standard_normal((8, 3, 7, 7))), 'bbox_pred_weight_test': mx.nd.array(rs.standard_normal((4, 16))),
           'bbox_pred_weight': mx.nd.zeros((4, 16))}
    aux = {'bn0_moving_var': mx.nd.array(rs.uniform(0.5, 1.5, 8))}
    prefix = str(tmp_path / 'SNIPER')
    mx.model.save_checkpoint(prefix, 7, None, arg, aux)
    a2, x2 = ref.utils.load_param(prefix, 7, convert=True, process=True)
    assert np.array_equal(a2['conv0_weight'].asnumpy(), arg['conv0_weight'].asnumpy())
    assert np.array_equal(x2['bn0_moving_var'].asnumpy(), aux['bn0_moving_var'].asnumpy())
    # process=True renames the *_test parameters over the training ones (utils.py:96-99)
    assert np.array_equal(a2['bbox_pred_weight'].asnumpy(), arg['bbox_pred_weight_test'].asnumpy()) and 'bbox_pred_weight_test' not in a2
