"""Symbolic graph capture with the subset of the ``mxnet.symbol`` API the reference touches
(SURVEY.md section 8(b)): the reference's ``symbols/faster/*.py`` build their networks through these
calls unchanged; ``sniper_amd.engine`` then lowers the captured graph onto the HIP kernels.

A Symbol is a list of (node, output-index) heads.  Nodes record op name, attributes and inputs;
learnable inputs that the caller does not pass are auto-created as Variables named
``<name>_<arg>`` exactly like MXNet does (``conv0_weight``, ``bn0_gamma``, ``bn0_moving_mean`` ...),
because the reference addresses parameters by those names (init_weight_rcnn, FIXED_PARAMS,
checkpoint_callback).
"""
import collections
import threading

import numpy as np

_tls = threading.local()


def _counter():
    if not hasattr(_tls, 'names'):
        _tls.names = collections.defaultdict(int)
    return _tls.names


def _auto_name(hint):
    c = _counter()
    name = '%s%d' % (hint, c[hint])
    c[hint] += 1
    return name


def _tup(v, n=2):
    if v is None:
        return None
    if isinstance(v, str):
        v = v.strip('()[] ')
        v = tuple(int(float(t)) for t in v.split(',') if t.strip() != '')
    if isinstance(v, (int, float, np.integer)):
        return (int(v),) * n
    return tuple(int(t) for t in v)


def _bool(v):
    if isinstance(v, str):
        return v.lower() in ('1', 'true')
    return bool(v)


class Node(object):
    __slots__ = ('op', 'name', 'attrs', 'inputs', 'num_outputs', 'is_aux', 'extra')

    def __init__(self, op, name, attrs=None, inputs=None, num_outputs=1):
        self.op = op            # None for a Variable
        self.name = name
        self.attrs = attrs or {}
        self.inputs = inputs or []   # list of (Node, out_idx)
        self.num_outputs = num_outputs
        self.is_aux = False
        self.extra = {}


# op -> (ordered input names, aux-state names, #outputs, output names)
_OPS = {
    'Convolution': (['data', 'weight', 'bias'], [], 1),
    'Deconvolution': (['data', 'weight', 'bias'], [], 1),
    'FullyConnected': (['data', 'weight', 'bias'], [], 1),
    'BatchNorm': (['data', 'gamma', 'beta'], ['moving_mean', 'moving_var'], 1),
    'DeformableConvolution': (['data', 'offset', 'weight', 'bias'], [], 1),
    'Activation': (['data'], [], 1),
    'Pooling': (['data'], [], 1),
    'Cast': (['data'], [], 1),
    'Reshape': (['data'], [], 1),
    'Flatten': (['data'], [], 1),
    'BlockGrad': (['data'], [], 1),
    'MakeLoss': (['data'], [], 1),
    'smooth_l1': (['data'], [], 1),
    'clip': (['data'], [], 1),
    'SoftmaxActivation': (['data'], [], 1),
    'SoftmaxOutput': (['data', 'label'], [], 1),
    'elemwise_add': (['lhs', 'rhs'], [], 1),
    '_plus': (['lhs', 'rhs'], [], 1),
    '_minus': (['lhs', 'rhs'], [], 1),
    '_mul': (['lhs', 'rhs'], [], 1),
    '_mul_scalar': (['data'], [], 1),
    '_plus_scalar': (['data'], [], 1),
    '_minus_scalar': (['data'], [], 1),
    'DeformablePSROIPooling': (['data', 'rois', 'trans'], [], 1),
    'MultiProposal': (['cls_prob', 'bbox_pred', 'im_info'], [], 2),
    'MultiProposalTarget': (['cls_prob', 'bbox_pred', 'im_info', 'gt_boxes', 'valid_ranges', 'crowd_boxes'], [], 4),
    # mask branch (symbols/faster/resnet_mx_101_e2e_mask.py:317-318,392-399)
    'MultiProposalTargetMask': (['cls_prob', 'bbox_pred', 'im_info', 'gt_boxes', 'valid_ranges'], [], 6),
    'MaskRcnnTarget': (['rois', 'mask_polys', 'mask_ids'], [], 2),
    'pick': (['data', 'index'], [], 1),
}
_OUT_NAMES = {
    'MultiProposal': ['output', 'score'],
    'MultiProposalTarget': ['output', 'label', 'bbox_target', 'bbox_weight'],
    'MultiProposalTargetMask': ['output', 'label', 'bbox_target', 'bbox_weight', 'mask_rois', 'mask_ids'],
    'MaskRcnnTarget': ['mask_targets', 'mask_cls'],
}
_PARAM_INPUTS = {'weight', 'bias', 'gamma', 'beta'}
_HINTS = {'_plus': '_plus', '_minus': '_minus', '_mul': '_mul', 'elemwise_add': 'elemwise_add'}


class Symbol(object):
    def __init__(self, heads):
        self._heads = list(heads)  # [(Node, idx)]

    # ---- composition -------------------------------------------------------------------------
    def __getitem__(self, i):
        if isinstance(i, str):
            names = self.list_outputs()
            i = names.index(i)
        if len(self._heads) == 1 and self._heads[0][0].num_outputs > 1:
            return Symbol([(self._heads[0][0], i)])
        return Symbol([self._heads[i]])

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def __len__(self):
        if len(self._heads) == 1:
            return self._heads[0][0].num_outputs
        return len(self._heads)

    def _bin(self, other, op, scalar_op, reverse=False):
        if isinstance(other, Symbol):
            a, b = (other, self) if reverse else (self, other)
            return _create(op, [a, b], {})
        attrs = {'scalar': float(other)}
        if reverse and scalar_op == '_minus_scalar':
            return _create('_plus_scalar', [_create('_mul_scalar', [self], {'scalar': -1.0})], attrs)
        return _create(scalar_op, [self], attrs)

    def __add__(self, o):
        return self._bin(o, '_plus', '_plus_scalar')

    __radd__ = __add__

    def __sub__(self, o):
        return self._bin(o, '_minus', '_minus_scalar')

    def __rsub__(self, o):
        return self._bin(o, '_minus', '_minus_scalar', reverse=True)

    def __mul__(self, o):
        return self._bin(o, '_mul', '_mul_scalar')

    __rmul__ = __mul__

    def __neg__(self):
        return self * -1.0

    # ---- introspection -----------------------------------------------------------------------
    @property
    def name(self):
        return self._heads[0][0].name if len(self._heads) == 1 else None

    def _topo(self):
        # (a Symbol is immutable once built -- every operator call returns a new one -- so the order is computed once; an executor
        # bind asks for it four times, and a test-time Module binds one executor per batch shape)
        cached = self.__dict__.get('_topo_cache')
        if cached is not None:
            return list(cached)
        order, seen = [], set()
        stack = [(h[0], False) for h in reversed(self._heads)]
        while stack:
            node, done = stack.pop()
            if done:
                order.append(node)
                continue
            if id(node) in seen:
                continue
            seen.add(id(node))
            stack.append((node, True))
            for inp, _ in reversed(node.inputs):
                if id(inp) not in seen:
                    stack.append((inp, False))
        self.__dict__['_topo_cache'] = tuple(order)
        return order

    def list_arguments(self):
        return [n.name for n in self._topo() if n.op is None and not n.is_aux]

    def list_auxiliary_states(self):
        return [n.name for n in self._topo() if n.op is None and n.is_aux]

    def list_outputs(self):
        out = []
        for node, idx in self._heads:
            if node.op is None:
                out.append(node.name)
            elif node.op in _OUT_NAMES:
                out.append('%s_%s' % (node.name, _OUT_NAMES[node.op][idx]))
            else:
                out.append('%s_output' % node.name)
        return out

    def get_internals(self):
        heads = []
        for n in self._topo():
            for i in range(n.num_outputs):
                heads.append((n, i))
        return Symbol(heads)

    def _set_attr(self, **kwargs):
        for node, _ in self._heads:
            node.extra.update(kwargs)

    def attr(self, key):
        return self._heads[0][0].extra.get(key)

    def infer_shape(self, **kwargs):
        from ..engine.shapes import infer_shapes
        shapes = infer_shapes(self, kwargs)
        args = [shapes[('var', n)] for n in self.list_arguments()]
        auxs = [shapes[('var', n)] for n in self.list_auxiliary_states()]
        outs = [shapes[(id(node), idx)] for node, idx in self._heads]
        return args, outs, auxs

    def infer_shape_partial(self, **kwargs):
        return self.infer_shape(**kwargs)

    def save(self, fname):
        import json
        nodes = self._topo()
        index = {id(n): i for i, n in enumerate(nodes)}
        js = {'nodes': [{'op': n.op or 'null', 'name': n.name,
                         'attrs': {k: str(v) for k, v in n.attrs.items()},
                         'inputs': [[index[id(i)], o] for i, o in n.inputs]} for n in nodes],
              'heads': [[index[id(n)], o] for n, o in self._heads]}
        with open(fname, 'w') as fh:
            json.dump(js, fh)

    def tojson(self):
        return repr(self.list_outputs())


def Variable(name, shape=None, dtype=None, lr_mult=None, wd_mult=None, init=None, **kwargs):
    node = Node(None, name)
    if shape is not None:
        node.extra['shape'] = tuple(shape)
    if lr_mult is not None:
        node.extra['lr_mult'] = float(lr_mult)
    if wd_mult is not None:
        node.extra['wd_mult'] = float(wd_mult)
    return Symbol([(node, 0)])


var = Variable


def Group(symbols):
    heads = []
    for s in symbols:
        heads.extend(s._heads)
    return Symbol(heads)


def _single(sym, what):
    if len(sym._heads) != 1:
        raise ValueError('%s: a grouped symbol cannot be used as an operator input' % what)
    return sym._heads[0]


def _create(op, pos_inputs, kwargs):
    if op not in _OPS and op != 'Concat' and op != 'Custom':
        raise NotImplementedError("operator %r is not part of the SNIPER hot path (sniper_amd.mx.symbol)" % op)
    kwargs = dict(kwargs)
    name = kwargs.pop('name', None)
    attr = kwargs.pop('attr', None)
    if op in ('Concat', 'Custom'):
        ins = list(pos_inputs)
        named = {}
        for k in list(kwargs):
            if isinstance(kwargs[k], Symbol):
                named[k] = kwargs.pop(k)
        if op == 'Custom':
            from . import operator as _operator
            prop = _operator.get_prop(kwargs.get('op_type'), kwargs)
            order = prop.list_arguments()
            ins = ins + [named[k] for k in order if k in named]
            nout = len(prop.list_outputs())
        else:
            nout = 1
        name = name or _auto_name(op.lower())
        node = Node(op, name, kwargs, [_single(s, op) for s in ins], nout)
        return Symbol([(node, 0)]) if nout == 1 else Symbol([(node, i) for i in range(nout)])
    in_names, aux_names, nout = _OPS[op]
    name = name or _auto_name(_HINTS.get(op, op.lower()))
    given = {}
    for k, s in zip(in_names, pos_inputs):
        given[k] = s
    for k in list(kwargs):
        if isinstance(kwargs[k], Symbol):
            given[k] = kwargs.pop(k)
    attrs = kwargs
    lr_mult = attrs.pop('lr_mult', None)
    wd_mult = attrs.pop('wd_mult', None)
    inputs = []
    for k in in_names:
        if k in given:
            inputs.append(_single(given[k], op))
            continue
        if k == 'bias' and _bool(attrs.get('no_bias', op == 'Deconvolution')):    # MXNet: Deconvolution defaults to no_bias
            continue
        if k in _PARAM_INPUTS:
            v = Variable('%s_%s' % (name, k))
            if lr_mult is not None:
                v._heads[0][0].extra['lr_mult'] = float(lr_mult)
            if wd_mult is not None:
                v._heads[0][0].extra['wd_mult'] = float(wd_mult)
            inputs.append(v._heads[0])
            continue
        if k in ('trans', 'crowd_boxes', 'gt_masks', 'index'):
            continue  # optional inputs
        raise ValueError('%s(%s): missing input %r' % (op, name, k))
    for k in aux_names:
        if k in given:
            a = _single(given[k], op)
        else:
            a = Variable('%s_%s' % (name, k))._heads[0]
        a[0].is_aux = True
        inputs.append(a)
    node = Node(op, name, attrs, inputs, nout)
    # remember which logical slot each input fills (optional inputs may be absent)
    slots = []
    for k in in_names:
        if k in given:
            slots.append(k)
        elif k == 'bias' and _bool(attrs.get('no_bias', op == 'Deconvolution')):
            continue
        elif k in _PARAM_INPUTS:
            slots.append(k)
    slots += aux_names
    node.extra['slots'] = slots
    if attr:
        node.extra.update(attr)
    if nout == 1:
        return Symbol([(node, 0)])
    return Symbol([(node, i) for i in range(nout)])


_POS_ATTRS = {'clip': ('a_min', 'a_max')}


def _make_op(op):
    def fn(*args, **kwargs):
        pos = [a for a in args if isinstance(a, Symbol)]
        # positional scalar attributes (mx.sym.clip(data, a_min, a_max), mobilenetv2_e2e.py:19)
        scalars = [a for a in args if not isinstance(a, Symbol)]
        for k, v in zip(_POS_ATTRS.get(op, ()), scalars):
            kwargs.setdefault(k, v)
        return _create(op, pos, kwargs)

    fn.__name__ = op
    return fn


def Concat(*args, **kwargs):
    return _create('Concat', [a for a in args if isinstance(a, Symbol)], kwargs)


def Custom(*args, **kwargs):
    return _create('Custom', [a for a in args if isinstance(a, Symbol)], kwargs)


def __getattr__(name):  # module-level: mx.sym.<AnyOp>
    if name.startswith('__'):
        raise AttributeError(name)
    if name in _OPS:
        return _make_op(name)
    raise AttributeError("mx.sym.%s is not provided by sniper_amd (not on the SNIPER hot path)" % name)


concat = Concat
