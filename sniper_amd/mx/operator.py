"""``mx.operator`` plugin API (lib/operator_py/box_annotator_ohem.py:19-120 is the exemplar):
CustomOp / CustomOpProp / register, with the same method contract (forward(is_train, req, in_data,
out_data, aux), backward(...), assign(dst, req, src); prop: list_arguments / list_outputs /
infer_shape / create_operator / declare_backward_dependency).  Custom operators run on host numpy
NDArrays; the only in-tree user is dead at runtime (`if False:` in resnext_mx_101.py:311-316) but must
import and register."""
from . import ndarray as nd

_REGISTRY = {}


class CustomOp(object):
    def forward(self, is_train, req, in_data, out_data, aux):
        raise NotImplementedError

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        raise NotImplementedError

    def assign(self, dst, req, src):
        if req == 'null':
            return
        if isinstance(src, nd.NDArray):
            src = src.asnumpy()
        if req in ('write', 'inplace'):
            dst[:] = src
        elif req == 'add':
            dst[:] = dst.asnumpy() + src


class CustomOpProp(object):
    def __init__(self, need_top_grad=False):
        self.need_top_grad_ = need_top_grad

    def infer_shape(self, in_shape):
        return in_shape, [in_shape[0]] * len(self.list_outputs()), []

    def infer_type(self, in_type):
        return in_type, [in_type[0]] * len(self.list_outputs()), [in_type[0]] * len(self.list_auxiliary_states())

    def list_outputs(self):
        return ['output']

    def list_arguments(self):
        return ['data']

    def list_auxiliary_states(self):
        return []

    def declare_backward_dependency(self, out_grad, in_data, out_data):
        deps = []
        if self.need_top_grad_:
            deps.extend(out_grad)
        deps.extend(in_data)
        deps.extend(out_data)
        return deps

    def create_operator(self, ctx, in_shapes, in_dtypes):
        return CustomOp()


def register(reg_name):
    def do_register(prop_cls):
        _REGISTRY[reg_name] = prop_cls
        return prop_cls

    return do_register


def get_prop(op_type, attrs):
    if op_type not in _REGISTRY:
        raise ValueError('custom operator %r is not registered' % op_type)
    kw = {k: str(v) for k, v in attrs.items() if k not in ('op_type', 'name')}
    return _REGISTRY[op_type](**kw)
