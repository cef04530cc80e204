"""Base class of the network definitions (contract of the reference's symbols/symbol.py:9-61)."""
import numpy as np


class Symbol(object):
    def __init__(self):
        self.sym = None
        self.arg_shape_dict = self.out_shape_dict = self.aux_shape_dict = None

    @property
    def symbol(self):
        return self.sym

    def get_bbox_param_names(self):
        raise NotImplementedError()

    def infer_shape(self, data_shape_dict):
        args, outs, auxs = self.sym.infer_shape(**data_shape_dict)
        self.arg_shape_dict = dict(zip(self.sym.list_arguments(), args))
        self.out_shape_dict = dict(zip(self.sym.list_outputs(), outs))
        self.aux_shape_dict = dict(zip(self.sym.list_auxiliary_states(), auxs))

    def get_msra_std(self, shape):
        return np.sqrt(2.0 / float(np.prod(shape[1:])))

    def check_parameter_shapes(self, arg_params, aux_params, data_shape_dict, is_train=True):
        for k in self.sym.list_arguments():
            if k in data_shape_dict or (not is_train and 'label' in k):
                continue
            assert k in arg_params, k + ' not initialized'
            assert tuple(arg_params[k].shape) == tuple(self.arg_shape_dict[k]), 'shape inconsistent for ' + k
        for k in self.sym.list_auxiliary_states():
            assert k in aux_params, k + ' not initialized'
            assert tuple(aux_params[k].shape) == tuple(self.aux_shape_dict[k]), 'shape inconsistent for ' + k
