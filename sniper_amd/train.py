"""Assemble the SNIPER training step (config -> roidb -> iterator -> symbol -> Module) the way
main_train.py:36-146 does, for benchmarks, the smoke test and the GPU tests."""
import numpy as np

import sniper_amd.mx as mx

from . import config as cfgmod
from .iterators.MNIteratorE2E import MNIteratorE2E
from .synthetic import make_roidb


def fixed_param_names(cfg, sym):
    """lib/train_utils/utils.py:103-117: every argument whose name contains a FIXED_PARAMS prefix."""
    out = []
    for name in sym.list_arguments():
        if any(p in name for p in (cfg.network.FIXED_PARAMS or [])):
            out.append(name)
    return out


def optimizer_params(cfg, lr_scheduler=None):
    """lib/train_utils/utils.py:13-42: fp16 folds the static loss scale into lr and wd."""
    tr = cfg.TRAIN
    if tr.fp16:
        return {'momentum': tr.momentum, 'wd': tr.wd * tr.scale, 'learning_rate': tr.lr / tr.scale, 'rescale_grad': 1.0,
                'multi_precision': True, 'clip_gradient': None, 'lr_scheduler': lr_scheduler}
    return {'momentum': tr.momentum, 'wd': tr.wd, 'learning_rate': tr.lr, 'rescale_grad': 1.0, 'clip_gradient': None,
            'lr_scheduler': lr_scheduler}


class Trainer(object):
    """R101 Faster-RCNN SNIPER training on synthetic COCO-shaped data (BASELINE C2/C3)."""

    def __init__(self, batch_images=20, n_images=64, seed=0, momentum=0.995, rank_local=True, n_proposals=0, cfg=None):
        self.cfg = cfg or cfgmod.res101_e2e(batch_images=batch_images)
        cfg = self.cfg
        cfg.TRAIN.USE_NEG_CHIPS = n_proposals > 0
        np.random.seed(seed)
        self.roidb = make_roidb(n_images, seed=seed, n_proposals=n_proposals, with_masks=bool(cfg.TRAIN.WITH_MASK))
        self.iter = MNIteratorE2E(self.roidb, cfg, batch_size=batch_images, nGPUs=1, im_source='synthetic')   # synthetic chips
        # main_train.py:83-84: the symbol class is named by the config
        import importlib
        name = cfg.get('symbol', 'resnet_mx_101_e2e')
        net_cls = getattr(importlib.import_module('sniper_amd.symbols.faster.' + name), name)
        self.net = net_cls(n_proposals=400, momentum=momentum)
        self.sym = self.net.get_symbol_rcnn(cfg)
        self.mod = mx.mod.Module(self.sym, context=[mx.gpu(0)], data_names=[k for k, _ in self.iter.provide_data_single],
                                 label_names=[k for k, _ in self.iter.provide_label_single],
                                 fixed_param_names=fixed_param_names(cfg, self.sym))
        self.mod.slice_inputs = not rank_local
        self.mod.bind(self.iter.provide_data, self.iter.provide_label, for_training=True)
        shape_dict = dict(self.iter.provide_data_single + self.iter.provide_label_single)
        self.net.infer_shape(shape_dict)
        arg, aux = {}, {}
        mx.random.seed(seed)
        self.net.init_weight_rcnn(cfg, arg, aux)   # heads N(0, .01), offsets zero; backbone: MSRA (no pretrained file here)
        self.mod.init_params(arg_params=arg, aux_params=aux, allow_missing=True)
        self.mod.init_optimizer(optimizer='sgd', optimizer_params=optimizer_params(cfg))
        self.batch = self.iter.batch

    def next_batch(self):
        try:
            self.batch = self.iter.next()
        except StopIteration:
            self.iter.reset()
            self.batch = self.iter.next()
        return self.batch

    def step(self, batch=None):
        b = batch if batch is not None else self.batch
        self.mod.forward_backward(b)
        self.mod.update()
        return self.mod.get_outputs()
