"""``mx.callback``, ``mx.lr_scheduler``, ``mx.model``, ``mx.random`` subsets."""
import logging
import os
import time

import numpy as np

from . import ndarray as nd


# ---- mx.lr_scheduler (lib/train_utils/lr_scheduler.py:10 subclasses LRScheduler, uses base_lr)
class LRScheduler(object):
    def __init__(self, base_lr=0.01):
        self.base_lr = base_lr

    def __call__(self, num_update):
        raise NotImplementedError


# ---- mx.callback (main_train.py:138-140)
class Speedometer(object):
    """Logs samples/s every `frequent` batches -- the only throughput instrument the reference has."""

    def __init__(self, batch_size, frequent=50, auto_reset=True):
        self.batch_size, self.frequent, self.auto_reset = batch_size, frequent, auto_reset
        self.init, self.tic, self.last_count = False, 0, 0

    def __call__(self, param):
        count = param.nbatch
        if self.last_count > count:
            self.init = False
        self.last_count = count
        if self.init:
            if count % self.frequent == 0:
                speed = self.frequent * self.batch_size / (time.time() - self.tic)
                if param.eval_metric is not None:
                    nv = param.eval_metric.get_name_value()
                    if self.auto_reset:
                        param.eval_metric.reset()
                    msg = 'Epoch[%d] Batch [%d]\tSpeed: %.2f samples/sec' % (param.epoch, count, speed)
                    msg += ''.join('\t%s=%f' % (n, v) for n, v in nv)
                    logging.info(msg)
                else:
                    logging.info('Iter[%d] Batch [%d]\tSpeed: %.2f samples/sec', param.epoch, count, speed)
                self.tic = time.time()
        else:
            self.init, self.tic = True, time.time()


def module_checkpoint(mod, prefix, period=1, save_optimizer_states=False):
    period = int(max(1, period))

    def _callback(iter_no, sym=None, arg=None, aux=None):
        if (iter_no + 1) % period == 0:
            mod.save_checkpoint(prefix, iter_no + 1, save_optimizer_states)

    return _callback


def do_checkpoint(prefix, period=1):
    def _callback(iter_no, sym, arg, aux):
        if (iter_no + 1) % max(1, int(period)) == 0:
            save_checkpoint(prefix, iter_no + 1, sym, arg, aux)

    return _callback


# ---- mx.model.save_checkpoint (resnet_mx_101_e2e.py:14)
def save_checkpoint(prefix, epoch, symbol, arg_params, aux_params):
    if symbol is not None:
        symbol.save('%s-symbol.json' % prefix)
    d = {('arg:%s' % k): v for k, v in arg_params.items()}
    d.update({('aux:%s' % k): v for k, v in aux_params.items()})
    name = '%s-%04d.params' % (prefix, epoch)
    tmp = '%s.tmp.%d' % (name, os.getpid())          # readers (and a second writer) never see a torn file
    nd.save(tmp, d)
    os.replace(tmp, name)
    logging.info('Saved checkpoint to "%s"', name)


def load_checkpoint(prefix, epoch):
    d = nd.load('%s-%04d.params' % (prefix, epoch))
    arg, aux = {}, {}
    for k, v in d.items():
        tp, name = k.split(':', 1)
        (arg if tp == 'arg' else aux)[name] = v
    return None, arg, aux


# ---- mx.random (resnet_mx_101_e2e.py:458 ...)
_rng = np.random.RandomState(0)


def seed(s):
    global _rng
    _rng = np.random.RandomState(s)


def normal(loc=0.0, scale=1.0, shape=(1,), ctx=None, dtype=None, **kw):
    return nd.NDArray(_rng.normal(loc, scale, size=shape).astype(np.float32))


def uniform(low=0.0, high=1.0, shape=(1,), ctx=None, dtype=None, **kw):
    return nd.NDArray(_rng.uniform(low, high, size=shape).astype(np.float32))
