"""CPU evaluation of a captured SNIPER symbol graph -- TEST INFRASTRUCTURE ONLY (never imported by sniper_amd/).

"Reference-semantics CPU" run of a whole network (BASELINE config C1 is the MobileNetV2 graph on
2 x 512 x 512 chips with reference ops on the CPU): every node of the graph the reference's symbol files build
(symbols/faster/mobilenetv2_e2e.py, resnet_mx_101_e2e.py) is evaluated in fp32 on the host --

  * standard operators (Convolution incl. grouped, BatchNorm, Activation, clip, Pooling, FullyConnected, Concat,
    Reshape, Cast, element-wise, SoftmaxActivation) by torch-CPU, with autograd for their gradients;
  * MXNet's loss operators by their documented gradient rules: SoftmaxOutput injects
    grad_scale * (p - onehot) / normaliser regardless of the incoming gradient (use_ignore / normalization='valid'),
    MakeLoss injects grad_scale, BlockGrad stops;
  * the fork-resident operators by the restatements of oracle/nn.py (MultiProposal(Target), DeformablePSROIPooling,
    DeformableConvolution sampling) -- **parity unpinned** like those: numpy loops, small sizes only.

`run(sym, params, aux, inputs)` -> (outputs, parameter gradients), all in the reference's layouts (OIHW weights).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import nn as onn


def _tup(v, n=2):
    if isinstance(v, str):
        v = tuple(int(float(t)) for t in v.strip('()[] ').split(',') if t.strip())
    if isinstance(v, (int, float, np.integer)):
        return (int(v),) * n
    return tuple(int(t) for t in v)


def _bool(v):
    return v in (True, 1, 'True', 'true', '1')


def _floats(v, default):
    if v is None:
        return tuple(default)
    if isinstance(v, str):
        return tuple(float(t) for t in v.strip('()[] ').split(',') if t.strip())
    return tuple(float(t) for t in v)


def _mx_reshape(src, spec):
    out, i = [], 0
    for s in spec:
        if s == 0:
            out.append(src[i]); i += 1
        elif s == -1:
            out.append(-1); i += 1
        else:
            out.append(int(s)); i += 1
    return tuple(out)


class _SoftmaxOutput(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, label, multi, use_ignore, ignore, grad_scale, norm):
        p = torch.softmax(x, 1 if multi else -1)
        ctx.save_for_backward(p, label)
        ctx.cfg = (multi, use_ignore, ignore, grad_scale, norm)
        return p

    @staticmethod
    def backward(ctx, _unused):
        p, label = ctx.saved_tensors
        multi, use_ignore, ignore, grad_scale, norm = ctx.cfg
        if multi:                                   # (B, K, ...) vs label (B, prod(...))
            B, K = p.shape[0], p.shape[1]
            pp = p.reshape(B, K, -1)
            lab = label.reshape(B, -1)
        else:                                       # (R, K) vs label (R,)
            pp = p.reshape(-1, p.shape[-1]).t().unsqueeze(0)
            lab = label.reshape(1, -1)
        valid = (lab != ignore) if use_ignore else torch.ones_like(lab, dtype=torch.bool)
        onehot = torch.zeros_like(pp).scatter_(1, lab.clamp(min=0).long().unsqueeze(1), 1.0)
        g = (pp - onehot) * valid.unsqueeze(1)
        denom = 1.0
        if norm == 'valid':
            denom = max(1, int(valid.sum()))
        elif norm == 'batch':
            denom = pp.shape[0] if multi else pp.shape[2]
        g = g * (grad_scale / denom)
        if not multi:
            g = g.squeeze(0).t()
        return g.reshape(p.shape), None, None, None, None, None, None


class _RoundF16(torch.autograd.Function):
    """value -> nearest fp16 (what the device stores for an activation); gradient passes straight through."""
    @staticmethod
    def forward(ctx, x):
        return x.half().float()

    @staticmethod
    def backward(ctx, g):
        return g


# operators whose output the engine keeps as an fp16 activation tensor (sniper_amd/engine/executor.py, _PRODUCES_ACT)
_F16_OUT = {'Convolution', 'FullyConnected', 'BatchNorm', 'Activation', 'Pooling', 'Concat', 'DeformableConvolution',
            'DeformablePSROIPooling', 'clip', '_plus', 'elemwise_add', 'Deconvolution', 'pick'}


class _MakeLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, grad_scale):
        ctx.gs = grad_scale
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return torch.full_like(g, ctx.gs), None


class _DPSROIPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, data, rois, trans, P, S, scale, tstd, G):
        d, r = data.detach().double().numpy(), rois.detach().numpy().astype(np.float32)
        t = None if trans is None else trans.detach().numpy().astype(np.float32)
        ctx.args = (d, r, t, P, S, scale, tstd, G)
        # the loop definition up to 64 RoIs, its sparse-operator form (pinned to it by tests) at BASELINE sizes
        fn = onn.dpsroi_pool if r.shape[0] <= 64 else onn.dpsroi_pool_fast
        return torch.from_numpy(fn(d, r, t, P, S, scale, tstd, G)).float()

    @staticmethod
    def backward(ctx, g):
        d, r, t, P, S, scale, tstd, G = ctx.args
        fn = onn.dpsroi_pool_backward if r.shape[0] <= 64 else onn.dpsroi_pool_backward_fast
        dd, dt = fn(g.double().numpy(), d, r, t, P, S, scale, tstd, G)
        return (torch.from_numpy(dd).float(), None, (None if dt is None else torch.from_numpy(dt).float()), None, None, None,
                None, None)


class _DeformIm2col(torch.autograd.Function):
    @staticmethod
    def forward(ctx, data, offset, k, stride, pad, dil, dg):
        d, o = data.detach().double().numpy(), offset.detach().double().numpy()
        ctx.args = (d, o, k, stride, pad, dil, dg)
        return torch.from_numpy(onn.deform_im2col(d, o, k[0], k[1], stride, pad, dil, dg)).float()   # (N,Ho,Wo,T,C)

    @staticmethod
    def backward(ctx, g):
        d, o, k, stride, pad, dil, dg = ctx.args
        dd, do = onn.deform_col2im(g.double().numpy(), d, o, k[0], k[1], stride, pad, dil, dg)
        return torch.from_numpy(dd).float(), torch.from_numpy(do).float(), None, None, None, None, None


def run(sym, params, aux, inputs, is_train=True, want_grads=True, overrides=None, fork_ops=True, fp16_storage=False,
        probe=None, force=None):
    """sym: sniper_amd.mx Symbol.  params / aux / inputs: {name: numpy array} in the reference's layouts.
    overrides: {(node name, output index): array} replaces that node output (used to compare the RoI heads on the
    very RoI set the device selected: a proposal whose score ties or whose IoU sits on the NMS threshold may
    legitimately differ between fp16 and fp32 features).  fork_ops=False skips the evaluation of an overridden
    MultiProposal(Target) altogether.  fp16_storage=True rounds every activation the engine stores in fp16 to fp16
    (arithmetic stays fp32): the ReLU / clip / max-pool decisions of a 50-layer network then coincide with the
    device's instead of flipping wherever a pre-activation lies within fp16 rounding of the kink, which is what
    an end-to-end GRADIENT comparison needs.  Returns (list of output arrays, {param name: gradient array or None})."""
    overrides = overrides or {}
    # (denormal gradients cost the CPU's slow path in every convolution they reach -- a third of the evaluation's time -- and are
    # 30 orders of magnitude below anything a comparison here resolves)
    torch.set_flush_denormal(True)
    # force: {node name: value} -- teacher forcing.  The node is evaluated by the oracle on its (forced) inputs, the
    # mismatch with the given value is recorded in run.local_err[name] (relative L2), and the given value replaces the
    # oracle's downstream while the GRADIENT still flows through the oracle's operator (y + (forced - y).detach()).
    # Every op is then compared on identical inputs and every ReLU / clip / max-pool takes the device's decisions, so
    # parameter gradients can be compared tightly instead of through the chaotic sensitivity of a random-init network.
    force = force or {}
    run.local_err = {}
    run.kept = {}          # node name -> (oracle value, forced value) for the names in run.keep (diagnostics: tools/parity_nodes.py)
    probes = {}     # probe: list of node names -> run() additionally returns {name: (value, gradient)} as a third result
    t = {k: torch.from_numpy(np.asarray(v, np.float32).copy()).requires_grad_(want_grads) for k, v in params.items()}
    auxt = {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in aux.items()}
    val = {}

    def get(n, i):
        return val[(id(n), i)]

    for node in sym._topo():
        a = node.attrs
        if node.op is None:
            if node.name in inputs:
                val[(id(node), 0)] = torch.from_numpy(np.asarray(inputs[node.name], np.float32))
            elif node.name in t:
                val[(id(node), 0)] = t[node.name]
            elif node.name in auxt:
                val[(id(node), 0)] = auxt[node.name]
            else:
                raise KeyError('no value for variable %s' % node.name)
            continue
        slots = node.extra.get('slots') or []
        ins = [get(n, i) for n, i in node.inputs]
        s = dict(zip(slots, ins))
        op = node.op
        if op == 'Convolution':
            y = F.conv2d(s['data'], s['weight'], s.get('bias'), _tup(a.get('stride', 1)), _tup(a.get('pad', 0)),
                         _tup(a.get('dilate', 1)), int(a.get('num_group', 1)))
        elif op == 'FullyConnected':
            x = s['data']
            y = F.linear(x.reshape(x.shape[0], -1), s['weight'], s.get('bias'))
        elif op == 'BatchNorm':
            x = s['data']
            g = torch.ones_like(s['beta']) if _bool(a.get('fix_gamma', True)) else s['gamma']
            eps = float(a.get('eps', 1e-3))
            if is_train and not _bool(a.get('use_global_stats', False)):
                y = F.batch_norm(x, None, None, g, s['beta'], True, 0.0, eps)
            else:
                sh = (1, -1, 1, 1)
                y = (x - s['moving_mean'].reshape(sh)) / torch.sqrt(s['moving_var'].reshape(sh) + eps) * g.reshape(sh) + \
                    s['beta'].reshape(sh)
        elif op == 'Activation':
            assert a.get('act_type') == 'relu'
            y = torch.relu(ins[0])
        elif op == 'clip':
            lo, hi = float(a['a_min']), float(a['a_max'])
            x = ins[0]
            y = x * ((x >= lo) & (x <= hi)).float() + (x.detach().clamp(lo, hi) - x.detach() * ((x >= lo) & (x <= hi)).float())
        elif op == 'Pooling':
            k = _tup(a['kernel'])
            if a.get('pool_type', 'max') == 'avg':
                assert _bool(a.get('global_pool', False)) or k == tuple(ins[0].shape[2:])
                y = ins[0].mean((2, 3), keepdim=True)
            else:
                y = F.max_pool2d(ins[0], k, _tup(a.get('stride', 1)), _tup(a.get('pad', 0)))
        elif op == 'Cast':
            y = ins[0]
        elif op == 'Concat':
            y = torch.cat(ins, int(a.get('dim', 1)))
        elif op in ('Reshape', 'Flatten'):
            x = ins[0]
            y = x.reshape(x.shape[0], -1) if op == 'Flatten' else x.reshape(_mx_reshape(tuple(x.shape), _tup(a['shape'], 1)))
        elif op in ('_plus', 'elemwise_add'):
            y = ins[0] + ins[1]
        elif op == '_minus':
            y = ins[0] - ins[1]
        elif op == '_mul':
            y = ins[0] * ins[1]
        elif op == '_mul_scalar':
            y = ins[0] * float(a['scalar'])
        elif op == '_plus_scalar':
            y = ins[0] + float(a['scalar'])
        elif op == '_minus_scalar':
            y = ins[0] - float(a['scalar'])
        elif op == 'BlockGrad':
            y = ins[0].detach()
        elif op == 'SoftmaxActivation':
            y = torch.softmax(ins[0], 1 if a.get('mode') == 'channel' else -1)
        elif op == 'SoftmaxOutput':
            y = _SoftmaxOutput.apply(ins[0], ins[1].detach(), _bool(a.get('multi_output', False)), _bool(a.get('use_ignore', False)),
                                     float(a.get('ignore_label', -1)), float(a.get('grad_scale', 1.0)), a.get('normalization', 'null'))
        elif op == 'smooth_l1':
            sg = float(a.get('scalar', 1.0))
            x = ins[0]
            y = torch.where(x.abs() < 1.0 / sg ** 2, 0.5 * (sg * x) ** 2, x.abs() - 0.5 / sg ** 2)
        elif op == 'MakeLoss':
            y = _MakeLoss.apply(ins[0], float(a.get('grad_scale', 1.0)))
        elif op == 'Deconvolution':
            y = F.conv_transpose2d(s['data'], s['weight'], s.get('bias'), _tup(a.get('stride', 1)), _tup(a.get('pad', 0)))
        elif op == 'pick':
            assert int(a.get('axis', -1)) == 1 and _bool(a.get('keepdims', False))
            x, idx = s['data'], s['index'].detach().long().clamp(0, s['data'].shape[1] - 1)
            y = x.gather(1, idx.reshape(-1, 1, 1, 1).expand(-1, 1, x.shape[2], x.shape[3]))
        elif op == 'MaskRcnnTarget':
            r, pl, ids = [s[k].detach().numpy() for k in ('rois', 'mask_polys', 'mask_ids')]
            tg, cl = onn.mask_rcnn_target(r, pl, ids, r.shape[0] // pl.shape[0], int(a.get('mask_size', 28)))
            y = (torch.from_numpy(tg), torch.from_numpy(cl))
        elif op in ('MultiProposal', 'MultiProposalTarget', 'MultiProposalTargetMask') and not fork_ops and (node.name, 0) in overrides:
            y = tuple(torch.from_numpy(np.asarray(overrides[(node.name, i)], np.float32)) for i in range(node.num_outputs))
        elif op in ('MultiProposal', 'MultiProposalTarget', 'MultiProposalTargetMask'):
            cls, box, info = [s[k].detach().numpy() for k in ('cls_prob', 'bbox_pred', 'im_info')]
            Fm = cls.shape[3]
            cls = cls.reshape(cls.shape[0], 2, -1, Fm)
            scales = _floats(a.get('scales'), (2, 4, 7, 10, 13, 16, 24))
            ratios = _floats(a.get('ratios'), (0.5, 1, 2))
            post = int(a.get('rpn_post_nms_top_n', 300))
            rois, scores, _ = onn.proposals(cls, box, info, int(a.get('feature_stride', 16)), scales, ratios,
                                            int(a.get('rpn_pre_nms_top_n', 6000)), post, float(a.get('threshold', 0.7)),
                                            float(a.get('rpn_min_size', 0)))
            if op == 'MultiProposal':
                y = (torch.from_numpy(rois), torch.from_numpy(scores))
            else:
                gtb, vrn = s['gt_boxes'].detach().numpy(), s['valid_ranges'].detach().numpy()
                lab, tgt, wgt = onn.proposal_targets(rois, gtb, vrn, post)
                y = (torch.from_numpy(rois), torch.from_numpy(lab), torch.from_numpy(tgt), torch.from_numpy(wgt))
                if op == 'MultiProposalTargetMask':
                    mr, mi = onn.mask_rois_select(rois, lab, onn.proposal_target_matches(rois, gtb, vrn, post), post,
                                                  int(a.get('num_mask_rois', 50)))
                    y = y + (torch.from_numpy(mr), torch.from_numpy(mi))
        elif op == 'DeformablePSROIPooling':
            no_trans = _bool(a.get('no_trans', False)) or 'trans' not in s
            y = _DPSROIPool.apply(s['data'], s['rois'].detach(), None if no_trans else s['trans'], int(a['pooled_size']),
                                  int(a.get('sample_per_part', 1)), float(a['spatial_scale']), float(a.get('trans_std', 0.0)),
                                  int(a.get('group_size', 1)))
        elif op == 'DeformableConvolution':
            k = _tup(a['kernel'])
            col = _DeformIm2col.apply(s['data'], s['offset'], k, _tup(a.get('stride', 1))[0], _tup(a.get('pad', 0))[0],
                                      _tup(a.get('dilate', 1))[0], int(a.get('num_deformable_group', 1)))
            N, Ho, Wo, T, C = col.shape
            w = s['weight']                                                    # (O, C, KH, KW) -> (O, T, C)
            y = torch.einsum('nhwtc,otc->nohw', col, w.permute(0, 2, 3, 1).reshape(w.shape[0], T, C))
            if 'bias' in s:
                y = y + s['bias'].reshape(1, -1, 1, 1)
        else:
            raise NotImplementedError('oracle.graph_cpu: operator %s (%s)' % (op, node.name))
        if fp16_storage and op in _F16_OUT and not isinstance(y, tuple):
            y = _RoundF16.apply(y)
        if node.name in force and not isinstance(y, tuple):
            fv = torch.from_numpy(np.asarray(force[node.name], np.float32)).reshape(y.shape)
            run.local_err[node.name] = float((y.detach() - fv).norm() / (fv.norm() + 1e-20))
            if node.name in getattr(run, 'keep', ()):
                run.kept[node.name] = (y.detach().numpy().copy(), fv.numpy().copy())
            y = y + (fv - y.detach())
        if isinstance(y, tuple):
            for i, yi in enumerate(y):
                val[(id(node), i)] = yi
        else:
            val[(id(node), 0)] = y
        if probe and node.name in probe and not isinstance(y, tuple) and val[(id(node), 0)].requires_grad:
            val[(id(node), 0)].retain_grad()
            probes[node.name] = val[(id(node), 0)]
        for i in range(node.num_outputs):
            if (node.name, i) in overrides:
                val[(id(node), i)] = torch.from_numpy(np.asarray(overrides[(node.name, i)], np.float32))
    outs = [get(n, i) for n, i in sym._heads]
    grads = {}
    if want_grads:
        diff = [o for o in outs if o.requires_grad]
        torch.autograd.backward(diff, [torch.ones_like(o) for o in diff])
        grads = {k: (v.grad.numpy() if v.grad is not None else None) for k, v in t.items()}
    if probe:
        return [o.detach().numpy() for o in outs], grads, {k: (v.detach().numpy(), None if v.grad is None else v.grad.numpy())
                                                            for k, v in probes.items()}
    return [o.detach().numpy() for o in outs], grads
