"""Shape inference over a captured symbol graph (logical shapes in the reference's NCHW order)."""
import numpy as np

from ..mx.symbol import _bool, _tup


def _conv_out(h, k, s, p, d):
    return (h + 2 * p - d * (k - 1) - 1) // s + 1


def slot_inputs(node):
    """{slot-name: (input-node, idx)} for operator nodes with named slots."""
    slots = node.extra.get('slots')
    if not slots:
        return {}
    return dict(zip(slots, node.inputs))


def mx_reshape(src, spec):
    """MXNet Reshape with the special codes 0 (copy), -1 (infer), -2 (copy rest), -3 (merge two),
    -4 (split) -- only 0 and -1 are used by the reference (resnet_mx_101_e2e.py:178,285,289,323)."""
    out, i = [], 0
    spec = list(spec)
    j = 0
    while j < len(spec):
        s = spec[j]
        if s == 0:
            out.append(src[i]); i += 1
        elif s == -1:
            out.append(-1); i += 1
        elif s == -2:
            out.extend(src[i:]); i = len(src)
        elif s == -3:
            out.append(src[i] * src[i + 1]); i += 2
        elif s == -4:
            a, b = spec[j + 1], spec[j + 2]
            if a == -1:
                a = src[i] // b
            if b == -1:
                b = src[i] // a
            out.extend([a, b]); i += 1; j += 2
        else:
            out.append(s); i += 1
        j += 1
    total = int(np.prod(src))
    if -1 in out:
        known = int(np.prod([o for o in out if o != -1]))
        out[out.index(-1)] = total // known
    assert int(np.prod(out)) == total, ('reshape', src, spec, out)
    return tuple(int(o) for o in out)


def infer_node(node, ins, shapes_of_var):
    """ins: list of input shapes (None when unknown, only allowed for parameters).  Returns
    (list of output shapes, {input position: inferred parameter shape})."""
    op, a = node.op, node.attrs
    slots = node.extra.get('slots') or []
    pos = {k: i for i, k in enumerate(slots)}
    par = {}
    if op in ('Convolution', 'DeformableConvolution', 'Deconvolution'):
        n, c, h, w = ins[pos['data']]
        k = _tup(a['kernel'])
        s = _tup(a.get('stride', (1, 1)))
        p = _tup(a.get('pad', (0, 0)))
        d = _tup(a.get('dilate', (1, 1)))
        nf = int(a['num_filter'])
        g = int(a.get('num_group', 1))
        if op == 'Deconvolution':
            par[pos['weight']] = (c, nf // g, k[0], k[1])
            out = (n, nf, (h - 1) * s[0] - 2 * p[0] + k[0], (w - 1) * s[1] - 2 * p[1] + k[1])
        else:
            par[pos['weight']] = (nf, c // g, k[0], k[1])
            out = (n, nf, _conv_out(h, k[0], s[0], p[0], d[0]), _conv_out(w, k[1], s[1], p[1], d[1]))
        if 'bias' in pos:
            par[pos['bias']] = (nf,)
        return [out], par
    if op == 'FullyConnected':
        x = ins[pos['data']]
        k = int(np.prod(x[1:]))
        nh = int(a['num_hidden'])
        par[pos['weight']] = (nh, k)
        if 'bias' in pos:
            par[pos['bias']] = (nh,)
        return [(x[0], nh)], par
    if op == 'BatchNorm':
        x = ins[pos['data']]
        for k in ('gamma', 'beta', 'moving_mean', 'moving_var'):
            par[pos[k]] = (x[1],)
        return [tuple(x)], par
    if op in ('Activation', 'Cast', 'BlockGrad', 'MakeLoss', 'smooth_l1', 'clip', 'SoftmaxActivation', '_mul_scalar',
              '_plus_scalar', '_minus_scalar'):
        return [tuple(ins[0])], par
    if op == 'SoftmaxOutput':
        x = ins[0]
        if ins[1] is None:
            par[1] = (x[0],) + tuple(x[2:]) if _bool(a.get('multi_output', False)) else (x[0],)
        return [tuple(x)], par
    if op in ('_plus', '_minus', '_mul', 'elemwise_add'):
        x, y = ins
        if x is None:
            x = y
        if y is None:
            par[1] = tuple(x)
        elif tuple(x) != tuple(y):
            # broadcasting of size-1 dims is all the reference needs (rpn weights * loss)
            assert len(x) == len(y) and all(p == q or p == 1 or q == 1 for p, q in zip(x, y)), (node.name, x, y)
            x = tuple(max(p, q) for p, q in zip(x, y))
        return [tuple(x)], par
    if op == 'Pooling':
        n, c, h, w = ins[0]
        if _bool(a.get('global_pool', False)):
            return [(n, c, 1, 1)], par
        k = _tup(a['kernel'])
        s = _tup(a.get('stride', (1, 1)))
        p = _tup(a.get('pad', (0, 0)))
        if a.get('pooling_convention', 'valid') == 'full':
            ho = int(np.ceil((h + 2 * p[0] - k[0]) / float(s[0]))) + 1
            wo = int(np.ceil((w + 2 * p[1] - k[1]) / float(s[1]))) + 1
        else:
            ho, wo = (h + 2 * p[0] - k[0]) // s[0] + 1, (w + 2 * p[1] - k[1]) // s[1] + 1
        return [(n, c, ho, wo)], par
    if op == 'Concat':
        dim = int(a.get('dim', 1))
        out = list(ins[0])
        out[dim] = sum(x[dim] for x in ins)
        return [tuple(out)], par
    if op == 'Reshape':
        return [mx_reshape(ins[0], _tup(a['shape'], 1))], par
    if op == 'Flatten':
        return [(ins[0][0], int(np.prod(ins[0][1:])))], par
    if op == 'DeformablePSROIPooling':
        x, r = ins[pos['data']], ins[pos['rois']]
        ps = int(a['pooled_size'])
        return [(r[0], int(a['output_dim']), ps, ps)], par
    if op == 'MultiProposal':
        cp = ins[0]
        post = int(a.get('rpn_post_nms_top_n', 300))
        # scores are 1-D: the reference's only consumer stacks `cscores[0:n, np.newaxis]` beside (n, 4) boxes (lib/inference.py:389-391)
        return [(cp[0] * post, 5), (cp[0] * post,)], par
    if op == 'MultiProposalTarget':
        cp = ins[0]
        post = int(a.get('rpn_post_nms_top_n', 300))
        r = cp[0] * post
        return [(r, 5), (r,), (r, 4), (r, 4)], par
    if op == 'MultiProposalTargetMask':     # + the mask RoIs of every chip and the gt_boxes row each one matched
        cp = ins[0]
        post, nm = int(a.get('rpn_post_nms_top_n', 300)), int(a.get('num_mask_rois', 50))
        r = cp[0] * post
        return [(r, 5), (r,), (r, 4), (r, 4), (cp[0] * nm, 5), (cp[0] * nm,)], par
    if op == 'MaskRcnnTarget':
        r = ins[pos['rois']]
        ms = int(a.get('mask_size', 28))
        return [(r[0], ms, ms), (r[0],)], par
    if op == 'pick':
        x = ins[pos['data']]
        axis = int(a.get('axis', -1)) % len(x)
        keep = _bool(a.get('keepdims', False))
        out = tuple(1 if i == axis else d for i, d in enumerate(x)) if keep else tuple(d for i, d in enumerate(x) if i != axis)
        if 'index' in pos and ins[pos['index']] is None:
            par[pos['index']] = tuple(d for i, d in enumerate(x) if i != axis)
        return [out], par
    if op == 'Custom':
        from ..mx import operator as _operator
        prop = _operator.get_prop(a.get('op_type'), a)
        outs = prop.infer_shape([list(s) for s in ins])[1]      # (in, out) or (in, out, aux): CustomOpProp allows both
        return [tuple(o) for o in outs], par
    raise NotImplementedError('shape inference for %s' % op)


def infer_shapes(sym, known):
    """known: {variable name: shape}.  Returns {('var', name): shape, (id(node), idx): shape}."""
    shapes = {}
    for node in sym._topo():
        if node.op is None:
            shp = known.get(node.name, node.extra.get('shape'))
            if shp is not None:
                shapes[('var', node.name)] = tuple(int(s) for s in shp)
                shapes[(id(node), 0)] = shapes[('var', node.name)]
            continue
        ins = [shapes.get((id(i), o)) for i, o in node.inputs]
        outs, par = infer_node(node, ins, shapes)
        for p, shp in par.items():
            inode, _ = node.inputs[p]
            if inode.op is None and ('var', inode.name) not in shapes:
                shapes[('var', inode.name)] = tuple(shp)
                shapes[(id(inode), 0)] = tuple(shp)
        missing = [node.inputs[i][0].name for i, s in enumerate(ins) if s is None and i not in par]
        if missing:
            raise ValueError('cannot infer shape of %s: unknown inputs %s' % (node.name, missing))
        for i, o in enumerate(outs):
            shapes[(id(node), i)] = tuple(o)
    return shapes
