"""The subset of the ``mxnet`` Python API the SNIPER reference imports ("Outer
boundary"), implemented over the HIP engine.  ``sniper_amd/compat`` exposes this package under the
name ``mxnet`` so that the reference's files run unmodified on top."""
import sys
import types

from . import io, metric, misc, ndarray, operator, symbol  # noqa: F401
from .misc import LRScheduler, Speedometer, do_checkpoint, module_checkpoint, save_checkpoint, load_checkpoint
from .module import Module
from .ndarray import Context, cpu, gpu

nd = ndarray
sym = symbol


def _ns(name, **members):
    m = types.ModuleType(name)
    m.__dict__.update(members)
    return m


mod = _ns('mxnet.mod', Module=Module)
module = mod
callback = _ns('mxnet.callback', Speedometer=Speedometer, module_checkpoint=module_checkpoint, do_checkpoint=do_checkpoint)
model = _ns('mxnet.model', save_checkpoint=save_checkpoint, load_checkpoint=load_checkpoint)
lr_scheduler = _ns('mxnet.lr_scheduler', LRScheduler=LRScheduler)
random = _ns('mxnet.random', normal=misc.normal, uniform=misc.uniform, seed=misc.seed)
contrib = _ns('mxnet.contrib', sym=symbol, symbol=symbol, nd=ndarray, ndarray=ndarray)
init = _ns('mxnet.init')
initializer = init
context = _ns('mxnet.context', Context=Context, cpu=cpu, gpu=gpu)
__version__ = '1.0.0-sniper_amd'


def alias_as(name='mxnet'):
    """Register this package and its sub-namespaces in sys.modules under `name`."""
    me = sys.modules[__name__]
    sys.modules[name] = me
    for sub, m in (('nd', ndarray), ('ndarray', ndarray), ('sym', symbol), ('symbol', symbol), ('io', io), ('metric', metric),
                   ('mod', mod), ('module', mod), ('callback', callback), ('model', model), ('lr_scheduler', lr_scheduler),
                   ('random', random), ('contrib', contrib), ('operator', operator), ('init', init), ('context', context)):
        sys.modules['%s.%s' % (name, sub)] = m
    sys.modules['%s.contrib.sym' % name] = symbol
    sys.modules['%s.contrib.symbol' % name] = symbol
