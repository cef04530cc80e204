"""Base class of the network definitions (contract of the reference's symbols/symbol.py:9-61: `symbol`, `infer_shape` filling the
three name -> shape dictionaries, `check_parameter_shapes`, `get_msra_std`)."""
import numpy as np


def _shape_table(names, shapes):
    return {n: tuple(s) for n, s in zip(names, shapes)}


class Symbol(object):
    sym = None
    arg_shape_dict = out_shape_dict = aux_shape_dict = None

    symbol = property(lambda self: self.sym)

    def get_bbox_param_names(self):
        raise NotImplementedError()

    def infer_shape(self, data_shape_dict):
        g = self.sym
        inferred = g.infer_shape(**data_shape_dict)
        listings = (g.list_arguments(), g.list_outputs(), g.list_auxiliary_states())
        self.arg_shape_dict, self.out_shape_dict, self.aux_shape_dict = (_shape_table(n, s) for n, s in zip(listings, inferred))

    def get_msra_std(self, shape):
        """sqrt(2 / fan_in), fan_in = every dimension of the weight but the first (:36-41)."""
        return np.sqrt(2.0 / float(np.prod(shape[1:])))

    def check_parameter_shapes(self, arg_params, aux_params, data_shape_dict, is_train=True):
        """Every argument that is not an input (nor, at test time, a label) and every auxiliary state has a value of the inferred shape."""
        def verify(names, given, want):
            for k in names:
                assert k in given, k + ' not initialized'
                assert tuple(given[k].shape) == tuple(want[k]), 'shape inconsistent for %s: inferred %s, provided %s' % (
                    k, tuple(want[k]), tuple(given[k].shape))
        skip = lambda k: k in data_shape_dict or (not is_train and 'label' in k)
        verify([k for k in self.sym.list_arguments() if not skip(k)], arg_params, self.arg_shape_dict)
        verify(self.sym.list_auxiliary_states(), aux_params, self.aux_shape_dict)
