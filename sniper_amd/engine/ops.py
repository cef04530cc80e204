"""Operator lowerings: one Step class per mx.sym operator the SNIPER graphs use.  Each step
allocates its outputs at construction (static shapes), launches HIP kernels in forward(), and in
backward() turns the gradients of its outputs into gradients of its inputs / parameters.

Semantics of the fork-resident operators are documented in DESIGN.md and restated in oracle/nn.py;
call sites: symbols/faster/resnet_mx_101_e2e.py, mobilenetv2_e2e.py.
"""
import os

import numpy as np
import torch

from .. import hip
from ..mx.symbol import _bool, _tup
from .executor import F16, F32, Val, _pad8
from .shapes import slot_inputs

REGISTRY = {}


def register(*names):
    def deco(cls):
        for n in names:
            REGISTRY[n] = cls
        return cls
    return deco


class Step(object):
    def __init__(self, ex, node, wants_f32):
        self.ex, self.node, self.wants_f32 = ex, node, wants_f32
        self.a = node.attrs
        self.ins = []
        for n, i in node.inputs:
            self.ins.append(ex.vals.get((id(n), i)))   # None for parameters / aux
        self.slots = slot_inputs(node)
        self.setup()

    # -- helpers
    def out_shape(self, i=0):
        return self.ex.shapes[(id(self.node), i)]

    def new_out(self, fmt, i=0, alloc=True):
        ex = self.ex
        v = Val('%s:%d' % (self.node.name, i), self.out_shape(i), fmt)
        if alloc:
            v.t = ex.act_empty(v.nhwc(), F16) if fmt == 'act' else ex.act_empty(v.shape, F32)
        ex.vals[(id(self.node), i)] = v
        v.producer = self
        return v

    def pname(self, slot):
        n, _ = self.slots[slot]
        return n.name

    def data_in(self, slot='data'):
        n, i = self.slots[slot]
        return self.ex.vals[(id(n), i)]

    def setup(self):
        raise NotImplementedError

    def forward(self):
        raise NotImplementedError

    def backward(self):
        pass

    def params_changed(self, only_trainable=False):
        pass

    def adopt_derived(self):
        """Executor.adopt_derived: True when this step needs nothing recomputed on a shape that shares its parameters"""
        return True

    def transpose_jobs(self, only_trainable=False):
        """extra (master, dst, O, T, I) transposed weight copies this step needs (Executor.transpose_jobs)"""
        return []


# ---------------------------------------------------------------------------------------------
# BatchNorm (+ fused ReLU when the only consumer is Activation(relu))
# ---------------------------------------------------------------------------------------------
@register('BatchNorm')
class BatchNormStep(Step):
    def setup(self):
        ex, a = self.ex, self.a
        self.x = self.data_in()
        C = self.x.shape[1]
        self.C = C
        self.eps = float(a.get('eps', 1e-3))
        self.momentum = float(a.get('momentum', 0.9))
        self.fix_gamma = _bool(a.get('fix_gamma', True))
        self.global_stats = _bool(a.get('use_global_stats', False))
        self.gamma = ex.register_param(self.pname('gamma'))
        self.beta = ex.register_param(self.pname('beta'))
        if self.fix_gamma:
            self.gamma.trainable = False
        if self.global_stats:
            # moving-statistics BN layers carry no gradient here: in every SNIPER config they are frozen
            # (network.FIXED_PARAMS) -- bn_data's beta, the one exception by name, is frozen too
            self.gamma.trainable = self.beta.trainable = False
        self.mean = ex.register_aux(self.pname('moving_mean'))
        self.var = ex.register_aux(self.pname('moving_var'))
        if self.pname('moving_var') not in ex.shared_names:      # (a shared statistic already holds the Module's values)
            ex.aux[self.pname('moving_var')].fill_(1.0)
        # (test time: one scale / shift per Module, shared by the executors of its batch shapes -- Executor.derived_buffer)
        self.scale = ex.derived_buffer(('bn_scale', self.node.name), (C,), F32)
        self.shift = ex.derived_buffer(('bn_shift', self.node.name), (C,), F32)
        self.save_mean, self.save_invstd = ex.empty((C,), F32), ex.empty((C,), F32)
        self.bws = None
        # image input (C <= 4): folded into the stem convolution's input packing
        self.is_stem = self.x.fmt == 'f32' and len(self.x.shape) == 4 and C <= 4
        cons = ex.consumers.get((id(self.node), 0), [])
        self.relu = (not self.is_stem and len(cons) == 1 and cons[0].op == 'Activation' and
                     cons[0].attrs.get('act_type') == 'relu')
        self.act = 1 if self.relu else 0      # fused activation code of the BN kernels: 0 none, 1 ReLU, 2 ReLU6
        if not self.is_stem and len(cons) == 1 and _is_relu6(cons[0]):
            self.act = 2
        # Test-time graphs: a BatchNorm that is the only reader of a convolution's output is a per-channel affine map (+ ReLU) of
        # it -- the convolution applies it itself (weights scaled by `scale`, bias = `shift`, ReLU in its epilogue) and this
        # step's output IS the convolution's output tensor: no separate pass over the activation (12 % of the AutoFocus pass).
        # Training graphs: the same holds for a FROZEN moving-statistics layer behind a frozen convolution (conv0 -> bn0 and
        # conv1 -> bn2, conv2 -> bn3 of the frozen stage-1 units, network.FIXED_PARAMS): constant affine map, no gradient on
        # either side, so the pass over the 128 x 128 / 256 x 256 maps (the largest tensors of the step) disappears.
        self.folded_into = None
        prod = None if self.is_stem else self.x.producer
        is_conv = prod is not None and type(prod).__name__ == 'ConvolutionStep' and self.x.fmt == 'act'
        frozen_pair = (ex.for_training and is_conv and self.global_stats and not prod.w.trainable
                       and (prod.b is None or not prod.b.trainable) and not prod.x.needs_grad
                       and os.environ.get('SNIPER_TRAIN_FOLD_BN', '1') != '0')
        if (is_conv and (not ex.for_training or frozen_pair) and self.act in (0, 1)
                and not prod.depthwise and not prod.out_f32
                and len(ex.consumers.get((id(prod.node), 0), [])) == 1 and (id(prod.node), 0) not in ex.head_keys
                and os.environ.get('SNIPER_INFER_FOLD_BN', '1') != '0'):
            self.folded_into = prod
            prod.fold_bn = self
        # Test-time graphs, second case: the BatchNorm (+ ReLU) that opens a pre-activation unit reads the residual SUM of the unit
        # before it -- two readers (this layer and the next add), so it cannot fold into the convolution that wrote the sum; that
        # convolution's epilogue has the sum in registers and writes act(scale * sum + shift) as a SECOND output straight into this
        # step's tensor (sn_conv_fwd_dual): the sn_bn_apply launch and its read + write of the whole tensor disappear (33 per R101
        # forward, 2 880 per 64-image AutoFocus pass).  SNIPER_INFER_DUAL_BN=0: the separate launch.
        self.dual_from = None
        if (not ex.for_training and self.folded_into is None and not self.is_stem and self.x.fmt == 'act' and self.act in (0, 1)
                and os.environ.get('SNIPER_INFER_DUAL_BN', '1') != '0'):
            src = self.x.producer
            conv = getattr(src, 'fused_conv', None) if type(src).__name__ == 'BinaryStep' else src
            if type(conv).__name__ == 'ConvolutionStep' and conv.request_dual(self):
                self.dual_from = conv
        if self.is_stem:
            self.y = self.new_out('f32', alloc=False)
            self.y.stem = (self.x, self.scale, self.shift)
        elif self.folded_into is not None:
            self.y = self.new_out('act', alloc=False)
            self.y.t = self.x.t
        else:
            self.y = self.new_out('act')
        self.y.needs_grad = ex.for_training and (self.x.needs_grad or self.gamma.trainable or self.beta.trainable) \
            and not self.is_stem
        self._global_ready = False
        # the one operator that consumes this layer's (activated) output, if there is exactly one: its data gradient is then
        # the complete dL/dy and may carry the backward reduction in its epilogue (ConvolutionStep.launch_dgrad)
        chain = cons
        if self.act and len(cons) == 1:
            chain = ex.consumers.get((id(cons[0]), 0), [])
        self.sole_consumer = chain[0] if len(chain) == 1 else None
        self.bwd_partials = None
        # batch statistics from the producing convolution's epilogue (plain conv -> BN, or the conv that absorbed the
        # residual add this BN reads): sn_conv_fwd_stats + sn_bn_finalize_blocks instead of sn_bn_stats + sn_bn_finalize
        self.stats_from = None
        if ex.for_training and not self.global_stats and not self.is_stem and self.x.fmt == 'act':
            prod = self.x.producer
            conv = getattr(prod, 'fused_conv', None) if type(prod).__name__ == 'BinaryStep' else prod
            if type(conv).__name__ == 'ConvolutionStep':
                got = conv.request_stats()
                if got is not None:
                    self.stats_from = got

    def params_changed(self, only_trainable=False):
        # after an optimizer step only trainable parameters moved; a moving-statistics layer with frozen
        # gamma/beta keeps its folded scale/shift
        if not (only_trainable and self.global_stats and not self.gamma.trainable and not self.beta.trainable):
            self._global_ready = False
            if self.global_stats or not self.ex.for_training:
                # fold the moving statistics NOW (persistent buffers): a forward pass that is a hipGraph replay never comes
                # back to Python, so a lazy recompute after set_params / a checkpoint load would keep the stale scale / shift.
                # A test-time executor normalises EVERY layer with the moving statistics, whatever use_global_stats says
                # (MobileNetV2's BatchNorm after the depthwise convolutions is neither global-stats nor folded).
                g = None if self.fix_gamma else self.gamma.master
                hip.call('sn_bn_global_scale_shift', g, self.beta.master, self.mean, self.var, self.C, self.eps, self.scale,
                         self.shift, hip.stream())
                self._global_ready = True
                if self.folded_into is not None:
                    self.folded_into.refold(self.scale, self.shift)

    def adopt_derived(self):
        # scale / shift are the Module's (derived_buffer) and hold the folded moving statistics; a folded convolution takes the
        # Module's folded weights
        if self.folded_into is not None:
            conv = self.folded_into
            ent = self.ex.fold_store.get(conv.w.name)
            if ent is None or conv.w.name not in self.ex.shared_names or ent[0].shape != conv.w.w16.shape:
                return False
            conv.wf, conv.bf = ent
        self._global_ready = True
        return True

    def _use_batch_stats(self):
        return self.ex.is_train and not self.global_stats

    def forward(self):
        ex = self.ex
        if self.folded_into is not None:
            return                       # the producing convolution wrote act(scale * conv + shift) into the shared tensor
        if self.dual_from is not None and not ex.is_train:
            return                       # ... or wrote it as its second output into this step's own tensor (sn_conv_fwd_dual)
        g = None if self.fix_gamma else self.gamma.master
        if not self._use_batch_stats():
            if not self._global_ready:
                hip.call('sn_bn_global_scale_shift', g, self.beta.master, self.mean, self.var, self.C, self.eps, self.scale,
                         self.shift, hip.stream())
                self._global_ready = True
            if self.is_stem:
                return
        if self.is_stem:
            raise NotImplementedError('batch-statistics BatchNorm on the raw image input')
        x = ex.as_act(self.x)
        n, h, w, c = self.x.nhwc()
        M = n * h * w
        if self._use_batch_stats():
            if self.bws is None:
                self.bws = ex.empty((hip.query('sn_bn_workspace_bytes', M, c),), torch.uint8)
            if self.stats_from is not None:
                # finalize + apply in one launch (the applying workgroups reduce the partials of their own channel slab)
                part, nblk = self.stats_from
                hip.call('sn_bn_apply_blocks', part, nblk, x, self.y.t, M, c, c, c, self.eps, self.momentum, g, self.beta.master,
                         self.mean, self.var, self.scale, self.shift, self.save_mean, self.save_invstd, self.act, hip.stream())
                return
            else:
                hip.call('sn_bn_stats', x, M, c, c, self.bws, hip.stream())
                hip.call('sn_bn_finalize', self.bws, M, c, self.eps, self.momentum, g, self.beta.master, self.mean,
                         self.var, self.scale, self.shift, self.save_mean, self.save_invstd, hip.stream())
        hip.call('sn_bn_apply', x, self.y.t, M, c, c, c, self.scale, self.shift, self.act, hip.stream())

    def backward(self):
        ex = self.ex
        if self.y.grad is None or not self.y.needs_grad:
            return
        if not self._use_batch_stats():
            if self.x.needs_grad or self.gamma.trainable or self.beta.trainable:
                raise NotImplementedError('gradient through a use_global_stats BatchNorm (%s)' % self.node.name)
            return
        n, h, w, c = self.x.nhwc()
        M = n * h * w
        x = ex.as_act(self.x)
        dx, acc = (None, None)
        if self.x.needs_grad:
            dx, acc = ex.grad_slot(self.x) if self.x.fmt == 'act' else (ex.empty(x.shape, F16), None)
        dg = self.gamma.grad if self.gamma.trainable else None
        db = self.beta.grad if self.beta.trainable else None
        if self.bwd_partials is not None:       # sum g, sum g*(x-mean) came out of the consumer's data-gradient epilogue
            part, nblk = self.bwd_partials
            self.bwd_partials = None
            hip.call('sn_bn_backward_blocks', part, nblk, self.y.grad, x, acc, dx, M, c, c, c, c, c, self.scale,
                     self.shift, self.save_mean, self.save_invstd, self.act, self.bws, dg, db, hip.stream())
        else:
            hip.call('sn_bn_backward', self.y.grad, x, acc, dx, M, c, c, c, c, c, self.scale, self.shift,
                     self.save_mean, self.save_invstd, self.act, self.bws, dg, db, hip.stream())
        if self.x.needs_grad and self.x.fmt != 'act':
            ex.add_grad(self.x, dx, 'act')
        self.y.grad = None


@register('Activation')
class ActivationStep(Step):
    def setup(self):
        ex = self.ex
        self.x = self.ins[0]
        if self.a.get('act_type') != 'relu':
            raise NotImplementedError('Activation %s' % self.a.get('act_type'))
        prod = self.node.inputs[0][0]
        self.fused = (prod.op == 'BatchNorm' and len(ex.consumers.get((id(prod), 0), [])) == 1 and self.x.fmt == 'act')
        if self.fused:
            ex.vals[(id(self.node), 0)] = self.x   # alias: the BN kernel already applied the ReLU
            self.y = self.x
        else:
            self.y = self.new_out('act')
            self.y.needs_grad = self.x.needs_grad

    def forward(self):
        if self.fused:
            return
        x = self.ex.as_act(self.x)
        n, h, w, c = self.x.nhwc()
        hip.call('sn_ew_f16', x, None, None, self.y.t, n * h * w, c, c, c, c, c, 0, hip.stream())

    def backward(self):
        if self.fused or self.y.grad is None or not self.x.needs_grad:
            return
        n, h, w, c = self.x.nhwc()
        if self.x.fmt == 'act':
            dx, acc = self.ex.grad_slot(self.x)
            hip.call('sn_ew_f16', self.y.grad, acc, self.y.t, dx, n * h * w, c, c, c, c, c, 2, hip.stream())
        else:
            tmp = self.ex.empty(self.y.t.shape, F16)
            hip.call('sn_ew_f16', self.y.grad, None, self.y.t, tmp, n * h * w, c, c, c, c, c, 2, hip.stream())
            self.ex.add_grad(self.x, tmp, 'act')
        self.y.grad = None


def _is_relu6(node):
    if node.op != 'clip':
        return False
    a = node.attrs
    return float(a.get('a_min', a.get('amin', 1))) == 0.0 and float(a.get('a_max', a.get('amax', 0))) == 6.0


@register('clip')
class ClipStep(Step):
    """mx.sym.clip(data, a_min, a_max) -- MobileNetV2's relu6 (mobilenetv2_e2e.py:18-19).  Behind a BatchNorm with
    no other consumer, clip(0, 6) is applied (and differentiated) by the BN kernels and this step is an alias."""

    def setup(self):
        ex, a = self.ex, self.a
        self.x = self.ins[0]
        self.lo = float(a.get('a_min', a.get('amin')))
        self.hi = float(a.get('a_max', a.get('amax')))
        prod = self.node.inputs[0][0]
        self.fused = (prod.op == 'BatchNorm' and len(ex.consumers.get((id(prod), 0), [])) == 1 and self.x.fmt == 'act' and
                      _is_relu6(self.node))
        if self.fused:
            ex.vals[(id(self.node), 0)] = self.x
            self.y = self.x
        else:
            self.y = self.new_out('act')
            self.y.needs_grad = self.x.needs_grad

    def forward(self):
        if self.fused:
            return
        x = self.ex.as_act(self.x)
        n, h, w, c = self.x.nhwc()
        hip.call('sn_clip_f16', x, None, None, self.y.t, n * h * w, c, c, c, c, c, self.lo, self.hi, 0, hip.stream())

    def backward(self):
        if self.fused or self.y.grad is None or not self.x.needs_grad:
            return
        n, h, w, c = self.x.nhwc()
        x = self.ex.as_act(self.x)
        if self.x.fmt == 'act':
            dx, acc = self.ex.grad_slot(self.x)
            hip.call('sn_clip_f16', self.y.grad, x, acc, dx, n * h * w, c, c, c, c, c, self.lo, self.hi, 1,
                     hip.stream())
        else:
            tmp = self.ex.empty(self.y.t.shape, F16)
            hip.call('sn_clip_f16', self.y.grad, x, None, tmp, n * h * w, c, c, c, c, c, self.lo, self.hi, 1, hip.stream())
            self.ex.add_grad(self.x, tmp, 'act')
        self.y.grad = None


# ---------------------------------------------------------------------------------------------
# Convolution / FullyConnected on the implicit-GEMM MFMA kernels
# ---------------------------------------------------------------------------------------------
def _wgrad(ex, dy, x, dw, N, H, W, C, x_ps, O, dy_ps, kh, kw, stride, pad, dil):
    """sn_conv_wgrad with the executor's shared split-K scratch (deterministic, atomic-free reduction) -- or, by default, queued
    for the executor's next batched launch (Executor.flush_wgrads)."""
    if ex.defer_wgrads:
        ex.queue_wgrad(dy, x, dw, N, H, W, C, x_ps, O, dy_ps, kh, kw, stride, pad, dil)
        return
    need = hip.query('sn_conv_wgrad_workspace_bytes', N, H, W, C, x_ps, O, dy_ps, kh, kw, stride, pad, dil)
    ws = ex.ws.get(need) if need else None
    hip.call('sn_conv_wgrad', dy, x, dw, N, H, W, C, x_ps, O, dy_ps, kh, kw, stride, pad, dil, ws, need, hip.stream())


def _bias_grad(ex, dy, db, rows, C, ld):
    """sn_bias_grad with the executor's scratch (ordered partial sums, no atomics); same stream as the weight gradient before it."""
    need = hip.query('sn_bias_grad_workspace_bytes', rows, C)
    ws = ex.ws.get(need) if need else None
    hip.call('sn_bias_grad', dy, db, rows, C, ld, 0, ws, need, hip.stream())


class _GemmLike(Step):
    """Shared by Convolution and FullyConnected.  Sub-classes fill geometry in setup_geom()."""

    def setup(self):
        ex = self.ex
        self.x = self.data_in()
        self.setup_geom()
        self.w = ex.register_param(self.pname('weight'), self.wkind, self.fc_in, need_wT=False)
        self.b = ex.register_param(self.pname('bias')) if 'bias' in self.slots else None
        self.out_f32 = self.wants_f32[0]
        self.y = self.new_out('f32' if self.out_f32 else 'act')
        wt = ex.for_training and (self.pname('weight') not in ex.fixed)
        self.y.needs_grad = ex.for_training and (self.x.needs_grad or wt)
        if self.x.needs_grad and ex.for_training and not getattr(self, 'depthwise', False):
            self.w.need_wT = True
        self.tmp_nhwc32 = None
        if self.out_f32 and self.Ho * self.Wo > 1:
            self.tmp_nhwc32 = ex.act_empty((self.N, self.Ho, self.Wo, self.O), F32)

    def refold(self, scale, shift):
        """Test-time fold of the BatchNorm that alone reads this convolution (BatchNormStep.setup): w' = scale[o] * w (from the
        fp32 master, one rounding to fp16), b' = scale * b + shift, into persistent buffers (a captured forward keeps their
        addresses)."""
        w = self.w.master.view(self.O, -1) * scale.view(-1, 1)
        b = shift if self.b is None else self.b.master * scale + shift
        if getattr(self, 'wf', None) is None:
            # one folded copy per Module: the executors of its other batch shapes fold the same masters with the same statistics
            ent = self.ex.fold_store.get(self.w.name) if self.w.name in self.ex.shared_names else None
            if ent is None or ent[0].shape != self.w.w16.shape:
                self.wf, self.bf = w.to(F16).view_as(self.w.w16).contiguous(), b.to(F32).contiguous().clone()
                self.ex.fold_store[self.w.name] = (self.wf, self.bf)
                return
            self.wf, self.bf = ent
        self.wf.copy_(w.view_as(self.wf))
        self.bf.copy_(b)

    def forward(self):
        ex = self.ex
        x = self.x_tensor()
        dst = self.tmp_nhwc32 if self.tmp_nhwc32 is not None else self.y.t
        bias = self.b.master if self.b is not None else None
        self.launch_fwd(x, dst, bias)
        if self.tmp_nhwc32 is not None:
            hw = self.Ho * self.Wo
            hip.call('sn_transpose_batched', dst, self.y.t, self.N, hw, self.O, hw * self.O, self.O * hw, self.O, hw, 1, 1,
                     hip.stream())

    def dy_act(self):
        """incoming gradient as channels-last fp16 with an 8-aligned channel pitch -> (tensor, pitch)"""
        ex = self.ex
        g = self.y.grad
        Op = _pad8(self.O)
        hw = self.Ho * self.Wo
        if self.y.fmt == 'act' and Op == self.O:
            return g, Op
        dy = ex.zeros((self.N, self.Ho, self.Wo, Op), F16) if Op != self.O else ex.empty((self.N, self.Ho, self.Wo, Op), F16)
        if self.y.fmt == 'act':
            hip.call('sn_copy2d', g, dy, self.N * hw, self.O, self.O, Op, 0, 0, hip.stream())
        elif hw == 1:
            hip.call('sn_copy2d', g, dy, self.N, self.O, self.O, Op, 1, 0, hip.stream())
        else:
            hip.call('sn_transpose_batched', g, dy, self.N, self.O, hw, self.O * hw, hw * Op, hw, Op, 1, 0, hip.stream())
        return dy, Op

    def backward(self):
        ex = self.ex
        if self.y.grad is None or not self.y.needs_grad:
            return
        dy, Op = self.dy_act()
        x = self.x_tensor()
        def param_grads():
            if self.w.trainable:
                self.launch_wgrad(dy, Op, x)
            if self.b is not None and self.b.trainable:
                _bias_grad(ex, dy, self.b.grad, self.N * self.Ho * self.Wo, self.O, Op)
        if self.w.trainable or (self.b is not None and self.b.trainable):
            ex.on_side(param_grads, keep=(dy, x))
        if self.x.needs_grad:
            if self.x.fmt == 'act':
                dx, acc = ex.grad_slot(self.x)
                self.launch_dgrad(dy, Op, acc, dx)
            else:
                dx = ex.empty(self.x_shape_nhwc(), F16)
                self.launch_dgrad(dy, Op, None, dx)
                ex.add_grad(self.x, dx, 'act')
        self.y.grad = None


@register('Convolution')
class ConvolutionStep(_GemmLike):
    def setup_geom(self):
        a = self.a
        self.k = _tup(a['kernel'])
        self.s = _tup(a.get('stride', (1, 1)))
        self.p = _tup(a.get('pad', (0, 0)))
        self.d = _tup(a.get('dilate', (1, 1)))
        assert self.s[0] == self.s[1] and self.p[0] == self.p[1] and self.d[0] == self.d[1], 'square geometry only'
        self.N, self.C, self.H, self.W = self.x.shape
        _, self.O, self.Ho, self.Wo = self.out_shape()
        g = int(a.get('num_group', 1))
        self.depthwise = g > 1
        if self.depthwise and not (g == self.C == self.O):
            raise NotImplementedError('grouped convolution with num_group=%d, %d -> %d channels (%s): only depthwise '
                                      '(num_group == channels, MobileNetV2) is built' % (g, self.C, self.O, self.node.name))
        self.wkind, self.fc_in = 'conv', None
        self.is_stem = self.C <= 4
        if self.is_stem:
            # the image convolution runs on a packed (N, Hp, Wp, 4) fp16 input; its weight lives in the matching
            # packed layout [O][KH][KWP*4] (zero padded), so forward, weight gradient and SGD all see one tensor
            ex = self.ex
            kh, kw = self.k
            self.KWP = (kw + 1) // 2 * 2
            self.Hp = (self.Ho - 1) * self.s[0] + kh
            self.Wp = ((self.Wo - 1) * self.s[1] + self.KWP + 1) // 2 * 2
            self.xp = ex.empty((self.N, self.Hp, self.Wp, 4), F16)
            self.wkind = 'stem'
            self.fc_in = (kh, kw, self.KWP)

    def x_tensor(self):
        return None if self.is_stem else self.ex.as_act(self.x)

    def x_shape_nhwc(self):
        return (self.N, self.H, self.W, self.C)

    def _stats_blocks(self):
        if self.is_stem or self.depthwise or self.out_f32 or os.environ.get('SNIPER_FUSE_BN_STATS', '1') == '0':
            return 0
        res = getattr(self, 'fused_residual', None)
        return hip.query('sn_conv_fwd_stats_blocks', self.N, self.H, self.W, self.C, self.C, self.O, self.O,
                         0 if res is None else self.O, self.k[0], self.k[1], self.s[0], self.p[0], self.d[0])

    def request_stats(self):
        """A batch-statistics BatchNorm reading this convolution's output (or the residual sum its epilogue writes) asks
        for the per-row-tile sums: -> (partials (blocks, 2, O) fp32, blocks) or None when the layer does not qualify."""
        if getattr(self, 'stats_buf', None) is None:
            nblk = self._stats_blocks()
            if nblk <= 0:
                return None
            self.stats_buf, self.stats_blocks = self.ex.empty((nblk, 2, self.O), F32), nblk
        return self.stats_buf, self.stats_blocks

    def request_dual(self, bn):
        """A test-time moving-statistics BatchNorm (+ ReLU) that reads this convolution's output -- or the residual sum its
        epilogue writes -- and cannot fold into it asks to be written as the epilogue's second output.  -> accepted?"""
        if (getattr(self, 'dual_bn', None) is not None or getattr(self, 'fold_bn', None) is not None or self.is_stem or
                self.depthwise or self.out_f32 or self.ex.for_training):
            return False
        res = getattr(self, 'fused_residual', None)
        if not hip.query('sn_conv_fwd_dual_ok', self.N, self.H, self.W, self.C, self.C, self.O, self.O, 0 if res is None else self.O,
                         self.k[0], self.k[1], self.s[0], self.p[0], self.d[0], self.O):
            return False
        self.dual_bn = bn
        return True

    def launch_fwd(self, x, dst, bias):
        ex = self.ex
        if self.is_stem:
            src, scale, shift = (self.x.stem if self.x.stem is not None else (self.x, None, None))
            hip.call('sn_pack_stem_input', src.t, self.xp, self.N, self.C, self.H, self.W, self.Hp, self.Wp, self.p[0], self.p[1],
                     scale, shift, hip.stream())
            fold = getattr(self, 'fold_bn', None)
            if fold is not None:     # frozen conv0 -> bn0 (-> relu0): the affine map rides in the weights / bias / epilogue (refold)
                if getattr(self, 'wf', None) is None:
                    raise RuntimeError('%s: folded BatchNorm weights missing (parameters were never set)' % self.node.name)
                hip.call('sn_conv_stem_fwd', self.xp, self.wf, self.bf, dst, self.N, self.Hp, self.Wp, self.Ho, self.Wo, self.O,
                         self.O, self.k[0], self.KWP, self.s[0], 1 if fold.act == 1 else 0, 0, hip.stream())
                return
            hip.call('sn_conv_stem_fwd', self.xp, self.w.w16, bias, dst, self.N, self.Hp, self.Wp, self.Ho, self.Wo, self.O,
                     self.O, self.k[0], self.KWP, self.s[0], 0, 1 if self.out_f32 else 0, hip.stream())
            return
        if self.depthwise:
            if bias is not None or self.out_f32:
                raise NotImplementedError('depthwise convolution with bias / fp32 output (%s)' % self.node.name)
            hip.call('sn_dwconv_fwd', x, self.w.w16, dst, self.N, self.H, self.W, self.C, self.C, self.O, self.k[0], self.k[1],
                     self.s[0], self.p[0], self.d[0], hip.stream())
            return
        res = getattr(self, 'fused_residual', None)
        if res is not None:      # y = conv(x) + residual written straight into the consuming add's tensor (BinaryStep)
            dst = self.fused_dst.t
        if getattr(self, 'stats_buf', None) is not None and self.ex.is_train:
            # the consuming BatchNorm's sum / sum of squares come out of the epilogue (no separate read pass)
            if self._stats_blocks() != self.stats_blocks:
                raise RuntimeError('%s: kernel selection changed after the statistics buffer was sized' % self.node.name)
            hip.call('sn_conv_fwd_stats', x, self.w.w16, bias, None if res is None else res.t, dst, self.N, self.H, self.W, self.C,
                     self.C, self.O, self.O, 0 if res is None else self.O, self.k[0], self.k[1], self.s[0], self.p[0], self.d[0], 0,
                     self.stats_buf, hip.stream())
            return
        fold = getattr(self, 'fold_bn', None)
        w, b, relu = self.w.w16, bias, 0
        if fold is not None:     # test-time: the BatchNorm (+ ReLU) reading this output is part of the epilogue (refold)
            if getattr(self, 'wf', None) is None:
                raise RuntimeError('%s: folded BatchNorm weights missing (parameters were never set)' % self.node.name)
            w, b, relu = self.wf, self.bf, 1 if fold.act == 1 else 0
        geom = (self.N, self.H, self.W, self.C, self.C, self.O, self.O, 0 if res is None else self.O, self.k[0], self.k[1], self.s[0],
                self.p[0], self.d[0])
        if not ex.for_training and not self.out_f32:
            # test-time launches with far fewer output tiles than CUs (batches of two FocusChips): contraction split over copies
            # of the tile grid (sn_conv_fwd_splitk; the query is 0 for every layer that would not be split)
            if getattr(self, '_splitk_bytes', None) is None:
                self._splitk_bytes = 0 if os.environ.get('SNIPER_CONV_SPLITK', '1') == '0' else \
                    int(hip.query('sn_conv_fwd_splitk_workspace_bytes', *geom))
            dual = getattr(self, 'dual_bn', None)
            if dual is not None and not ex.is_train:
                # the next unit's BatchNorm + ReLU as the epilogue's second output (BatchNormStep.setup), split-K or not
                if not dual._global_ready:
                    hip.call('sn_bn_global_scale_shift', None if dual.fix_gamma else dual.gamma.master, dual.beta.master, dual.mean,
                             dual.var, dual.C, dual.eps, dual.scale, dual.shift, hip.stream())
                    dual._global_ready = True
                hip.call('sn_conv_fwd_dual', x, w, b, None if res is None else res.t, dst, *geom, relu, dual.y.t, self.O, dual.scale,
                         dual.shift, 1 if dual.act == 1 else 0, ex.ws.get(self._splitk_bytes) if self._splitk_bytes else None,
                         self._splitk_bytes, hip.stream())
                return
            if self._splitk_bytes:
                hip.call('sn_conv_fwd_splitk', x, w, b, None if res is None else res.t, dst, *geom, relu, ex.ws.get(self._splitk_bytes),
                         self._splitk_bytes, hip.stream())
                return
        hip.call('sn_conv_fwd', x, w, b, None if res is None else res.t, dst, *geom, relu, 1 if self.out_f32 else 0, hip.stream())

    def launch_dgrad(self, dy, Op, acc, dx):
        if self.depthwise:
            hip.call('sn_dwconv_dgrad', dy, self.w.w16, acc, dx, self.N, self.H, self.W, self.C, Op, self.C, self.C, self.k[0],
                     self.k[1], self.s[0], self.p[0], self.d[0], hip.stream())
            return
        bn = self._bn_below(acc)
        if bn is not None:
            nblk = hip.query('sn_conv_dgrad_bn_blocks', self.N, self.H, self.W, self.C, self.C, Op, Op, 0, self.k[0], self.k[1],
                             self.s[0], self.p[0], self.d[0])
            if nblk > 0:
                if getattr(self, 'bnb_buf', None) is None or self.bnb_buf.shape[0] != nblk:
                    self.bnb_buf = self.ex.empty((nblk, 2, self.C), F32)
                hip.call('sn_conv_dgrad_bn', dy, self.w.wT16, None, dx, self.N, self.H, self.W, self.C, self.C, Op, Op, 0, self.k[0],
                         self.k[1], self.s[0], self.p[0], self.d[0], self.ex.as_act(bn.x), self.C, bn.scale, bn.shift, bn.save_mean,
                         bn.act, self.bnb_buf, hip.stream())
                bn.bwd_partials = (self.bnb_buf, nblk)
                return
        hip.call('sn_conv_dgrad', dy, self.w.wT16, acc, dx, self.N, self.H, self.W, self.C, self.C, Op, Op, self.C,
                 self.k[0], self.k[1], self.s[0], self.p[0], self.d[0], 0, hip.stream())

    def _bn_below(self, acc):
        """The batch-statistics BatchNorm whose (activated) output is this convolution's input and nobody else's: the data
        gradient written here is then its complete dL/dy, and the epilogue can carry the backward reduction.
        On by default since the epilogue reads the BatchNorm input 16 bytes per lane, prefetched before the K loop, with the
        per-channel constants hoisted (27.29 -> 26.98 ms per step, same box; with the earlier 8-byte epilogue it cost 1.5 %).
        SNIPER_FUSE_BN_BWD=0 restores the separate reduction pass."""
        if acc is not None or os.environ.get('SNIPER_FUSE_BN_BWD', '1') != '1' or self.x.fmt != 'act':
            return None
        bn = self.x.producer
        if type(bn).__name__ != 'BatchNormStep' or bn.sole_consumer is not self.node or bn.global_stats or bn.is_stem:
            return None
        if not (self.ex.is_train and bn.y.needs_grad and bn.x.fmt == 'act'):
            return None
        return bn

    def launch_wgrad(self, dy, Op, x):
        if self.is_stem:
            # the packed input self.xp still holds this step's image batch (written by launch_fwd)
            need = hip.query('sn_conv_stem_wgrad_workspace_bytes', self.N, self.Ho, self.Wo, self.O, self.k[0], self.KWP)
            ws = self.ex.ws.get(need) if need else None
            hip.call('sn_conv_stem_wgrad', dy, self.xp, self.w.grad, self.N, self.Hp, self.Wp, self.Ho, self.Wo, self.O, Op,
                     self.k[0], self.KWP, self.s[0], ws, need, hip.stream())
            return
        if self.depthwise:
            need = hip.query('sn_dwconv_wgrad_workspace_bytes', self.N, self.H, self.W, self.C, self.k[0], self.k[1], self.s[0], self.p[0], self.d[0])
            hip.call('sn_dwconv_wgrad', dy, x, self.w.grad, self.N, self.H, self.W, self.C, Op, self.C, self.k[0], self.k[1],
                     self.s[0], self.p[0], self.d[0], self.ex.ws.get(need), need, hip.stream())
            return
        _wgrad(self.ex, dy, x, self.w.grad, self.N, self.H, self.W, self.C, self.C, self.O, Op, self.k[0], self.k[1],
               self.s[0], self.p[0], self.d[0])


def _fwd_splitk(step, x, w, bias, dst, geom):
    """Test-time GEMM-shaped launches with a handful of output tiles (FullyConnected over a few hundred RoIs, the deformable
    convolution's column GEMM on a 2-chip batch): the contraction split over copies of the tile grid, as ConvolutionStep.launch_fwd
    does for its own layers.  True when the split launch was made (sn_conv_fwd_splitk_workspace_bytes is 0 for everything else)."""
    ex = step.ex
    if ex.for_training or ex.is_train:
        return False
    nbytes = getattr(step, '_splitk_bytes', None)
    if nbytes is None:
        nbytes = step._splitk_bytes = 0 if '0' in (os.environ.get('SNIPER_CONV_SPLITK', '1'), os.environ.get('SNIPER_FC_SPLITK', '1')) else \
            int(hip.query('sn_conv_fwd_splitk_workspace_bytes', *geom))
    if not nbytes:
        return False
    if getattr(step, 'out_f32', False):
        N, H, W, C, ips, O, ops_, _rps, KH, KW, st, pad, dil = geom
        hip.call('sn_conv_fwd_splitk_f32', x, w, bias, dst, N, H, W, C, ips, O, ops_, KH, KW, st, pad, dil, 0, ex.ws.get(nbytes), nbytes,
                 hip.stream())
    else:
        hip.call('sn_conv_fwd_splitk', x, w, bias, None, dst, *geom, 0, ex.ws.get(nbytes), nbytes, hip.stream())
    return True


@register('FullyConnected')
class FullyConnectedStep(_GemmLike):
    def setup_geom(self):
        xs = self.x.shape
        self.N = xs[0]
        self.H = self.W = self.Ho = self.Wo = 1
        self.C = int(np.prod(xs[1:]))
        self.O = int(self.a['num_hidden'])
        self.wkind = 'fc'
        # a 4-D channels-last input is consumed in (h, w, c) order: the weight is permuted once instead
        self.fc_in = (xs[1], xs[2], xs[3]) if len(xs) == 4 and xs[2] * xs[3] > 1 else None
        self.k, self.s, self.p, self.d = (1, 1), (1, 1), (0, 0), (1, 1)
        self.is_stem = False

    def x_tensor(self):
        return self.ex.as_act(self.x)

    def x_shape_nhwc(self):
        return self.x.nhwc()

    def launch_fwd(self, x, dst, bias):
        if _fwd_splitk(self, x, self.w.w16, bias, dst, (self.N, 1, 1, self.C, self.C, self.O, self.O, 0, 1, 1, 1, 0, 1)):
            return
        hip.call('sn_conv_fwd', x, self.w.w16, bias, None, dst, self.N, 1, 1, self.C, self.C, self.O, self.O, 0, 1, 1, 1, 0, 1, 0,
                 1 if self.out_f32 else 0, hip.stream())

    def launch_dgrad(self, dy, Op, acc, dx):
        hip.call('sn_conv_dgrad', dy, self.w.wT16, acc, dx, self.N, 1, 1, self.C, self.C, Op, Op, self.C, 1, 1, 1, 0, 1, 0,
                 hip.stream())

    def launch_wgrad(self, dy, Op, x):
        _wgrad(self.ex, dy, x, self.w.grad, self.N, 1, 1, self.C, self.C, self.O, Op, 1, 1, 1, 0, 1)


@register('DeformableConvolution')
class DeformableConvolutionStep(Step):
    """DCN v1: bilinear gather into a (M, T*C) column buffer + 1x1 GEMM (call site :124-128)."""

    def setup(self):
        ex, a = self.ex, self.a
        self.x = self.data_in()
        self.off = self.data_in('offset')
        self.k, self.s = _tup(a['kernel']), _tup(a.get('stride', (1, 1)))
        self.p, self.d = _tup(a.get('pad', (0, 0))), _tup(a.get('dilate', (1, 1)))
        self.dg = int(a.get('num_deformable_group', 1))
        self.N, self.C, self.H, self.W = self.x.shape
        _, self.O, self.Ho, self.Wo = self.out_shape()
        self.T = self.k[0] * self.k[1]
        self.w = ex.register_param(self.pname('weight'), 'conv', None, need_wT=ex.for_training)
        self.b = ex.register_param(self.pname('bias')) if 'bias' in self.slots else None
        self.y = self.new_out('act')
        self.y.needs_grad = ex.for_training
        self.col = ex.act_empty((self.N * self.Ho * self.Wo, self.T * self.C), F16)
        self.wT_flat = None

    def transpose_jobs(self, only_trainable=False):
        """the deformable convolution's data gradient is a 1x1 GEMM over the (tap, channel) column: W^T as [T*C][O]"""
        if not self.ex.for_training or (only_trainable and not self.w.trainable):
            return []
        if self.wT_flat is None:
            self.wT_flat = self.ex.zeros((self.T * self.C, 1, _pad8(self.O)), F16)
        return [(self.w.master, self.wT_flat, self.O, 1, self.T * self.C)]

    def forward(self):
        ex = self.ex
        x, off = ex.as_act(self.x), ex.as_act(self.off)
        oc = self.off.shape[1]
        hip.call('sn_deform_im2col', x, off, self.col, self.N, self.H, self.W, self.C, self.k[0], self.k[1], self.s[0], self.p[0],
                 self.d[0], self.dg, oc, 0, hip.stream())
        M, K = self.col.shape
        bias = self.b.master if self.b is not None else None
        if _fwd_splitk(self, self.col, self.w.w16, bias, self.y.t, (M, 1, 1, K, K, self.O, self.O, 0, 1, 1, 1, 0, 1)):
            return
        hip.call('sn_conv_fwd', self.col, self.w.w16, bias, None, self.y.t, M, 1, 1, K, K, self.O, self.O, 0, 1, 1, 1, 0, 1, 0, 0,
                 hip.stream())

    def backward(self):
        ex = self.ex
        if self.y.grad is None:
            return
        dy = self.y.grad
        M, K = self.col.shape
        Op = _pad8(self.O)
        assert Op == self.O
        def param_grads():
            if self.w.trainable:
                _wgrad(ex, dy, self.col, self.w.grad, M, 1, 1, K, K, self.O, Op, 1, 1, 1, 0, 1)
            if self.b is not None and self.b.trainable:
                _bias_grad(ex, dy, self.b.grad, M, self.O, Op)
        if self.w.trainable or (self.b is not None and self.b.trainable):
            ex.on_side(param_grads, keep=(dy,))
        if self.x.needs_grad or self.off.needs_grad:
            dcol = ex.empty((M, K), F16)
            hip.call('sn_conv_dgrad', dy, self.wT_flat, None, dcol, M, 1, 1, K, K, Op, Op, K, 1, 1, 1, 0, 1, 0, hip.stream())
            oc = self.off.shape[1]
            d16 = ex.empty((self.N, self.H, self.W, self.C), F16) if self.x.needs_grad else None
            d_off = ex.empty((self.N, self.Ho, self.Wo, oc), F16) if self.off.needs_grad else None
            if getattr(self, 'dws', None) is None:
                self.dws = ex.zeros((16,), torch.uint8)      # max |offset| of the launch: prunes the data gradient's scan
            hip.call('sn_deform_col2im', dcol, ex.as_act(self.x), ex.as_act(self.off), d16, 0, d_off, self.N, self.H, self.W,
                     self.C, self.k[0], self.k[1], self.s[0], self.p[0], self.d[0], self.dg, oc, 0, self.dws, hip.stream())
            if self.x.needs_grad:
                ex.add_grad(self.x, d16, 'act')
            if self.off.needs_grad:
                ex.add_grad(self.off, d_off, 'act')
        self.y.grad = None


# ---------------------------------------------------------------------------------------------
# structure ops
# ---------------------------------------------------------------------------------------------
@register('Pooling')
class PoolingStep(Step):
    """max pooling (the stem) and global average pooling (the vote of the position-sensitive R-FCN head,
    BASELINE config C4: act (N,H,W,C) -> f32 (N,C,1,1))."""

    def setup(self):
        self.x = self.ins[0]
        a = self.a
        self.kind = a.get('pool_type', 'max')
        self.glob = _bool(a.get('global_pool', False))
        n, c, h, w = self.x.shape
        if self.kind == 'avg' and (self.glob or _tup(a['kernel']) == (h, w) and _tup(a.get('pad', (0, 0))) == (0, 0)):
            self.kind = 'gavg'
            self.y = self.new_out('f32')
            self.y.needs_grad = self.x.needs_grad
            return
        if self.kind != 'max' or self.glob:
            raise NotImplementedError('Pooling %s (%s)' % (a.get('pool_type'), self.node.name))
        self.k, self.s, self.p = _tup(a['kernel']), _tup(a.get('stride', (1, 1))), _tup(a.get('pad', (0, 0)))
        self.y = self.new_out('act')
        self.y.needs_grad = self.x.needs_grad

    def forward(self):
        n, h, w, c = self.x.nhwc()
        if self.kind == 'gavg':
            hip.call('sn_avgpool_global_fwd', self.ex.as_act(self.x), self.y.t, n, h * w, c, hip.stream())
            return
        hip.call('sn_maxpool_fwd', self.ex.as_act(self.x), self.y.t, n, h, w, c, self.k[0], self.s[0], self.p[0], hip.stream())

    def backward(self):
        if self.y.grad is None or not self.x.needs_grad:
            return
        if self.kind != 'gavg':
            raise NotImplementedError('max-pool backward (the stem is frozen in every SNIPER config)')
        n, h, w, c = self.x.nhwc()
        dx = self.ex.empty((n, h, w, c), F16)
        hip.call('sn_avgpool_global_bwd', self.y.grad, dx, n, h * w, c, hip.stream())
        self.ex.add_grad(self.x, dx, 'act')
        self.y.grad = None


@register('Cast')
class CastStep(Step):
    """fp16 <-> fp32 casts of the reference graph (resnet_mx_101_e2e.py:250-252,405-406) are
    no-ops here: activations are fp16 channels-last on both sides, consumers convert on demand."""

    def setup(self):
        self.x = self.ins[0]
        self.ex.vals[(id(self.node), 0)] = self.x

    def forward(self):
        pass


@register('BlockGrad')
class BlockGradStep(Step):
    def setup(self):
        self.x = self.ins[0]
        self.y = self.new_out('f32', alloc=False)

    def forward(self):
        self.y.t = self.ex.as_f32(self.x)


@register('Concat')
class ConcatStep(Step):
    def setup(self):
        if int(self.a.get('dim', 1)) != 1:
            raise NotImplementedError('Concat along dim %s' % self.a.get('dim'))
        self.y = self.new_out('act')
        self.y.needs_grad = any(v.needs_grad for v in self.ins)

    def forward(self):
        n, h, w, ctot = self.y.nhwc()
        off = 0
        flat = self.y.t.view(-1)
        for v in self.ins:
            c = v.nhwc()[3]
            hip.call('sn_copy2d', self.ex.as_act(v), flat[off:], n * h * w, c, c, ctot, 0, 0, hip.stream())
            off += c

    def backward(self):
        if self.y.grad is None:
            return
        n, h, w, ctot = self.y.nhwc()
        off = 0
        flat = self.y.grad.view(-1)
        for v in self.ins:
            c = v.nhwc()[3]
            if v.needs_grad:
                g = self.ex.empty((n, h, w, c), F16)
                hip.call('sn_copy2d', flat[off:], g, n * h * w, c, ctot, c, 0, 0, hip.stream())
                self.ex.add_grad(v, g, 'act')
            off += c
        self.y.grad = None


@register('Reshape', 'Flatten')
class ReshapeStep(Step):
    def setup(self):
        self.x = self.ins[0]
        self.y = self.new_out('f32', alloc=False)
        self.y.needs_grad = self.x.needs_grad

    def forward(self):
        self.y.t = self.ex.as_f32(self.x).view(self.y.shape)

    def backward(self):
        if self.y.grad is None:
            return
        if self.x.needs_grad:
            self.ex.add_grad(self.x, self.y.grad.view(self.x.shape), 'f32')
        self.y.grad = None


@register('_plus', 'elemwise_add', '_minus', '_mul')
class BinaryStep(Step):
    def setup(self):
        a, b = self.ins
        self.lhs, self.rhs = a, b
        self.act = (a.fmt == 'act' and b.fmt == 'act' and not self.wants_f32[0] and self.node.op in ('_plus', 'elemwise_add'))
        self.y = self.new_out('act' if self.act else 'f32')
        self.y.needs_grad = a.needs_grad or b.needs_grad
        if not self.act and a.shape != b.shape:
            raise NotImplementedError('broadcasting in %s (%s)' % (self.node.op, self.node.name))
        self.code = {'_plus': 1, 'elemwise_add': 1, '_minus': 0, '_mul': 2}[self.node.op]
        # residual fusion: `conv(x) + shortcut` (every ResNet bottleneck, resnet_mx_101_e2e.py:66-69) is computed by the
        # convolution's epilogue.  The operand produced LAST is the one that can absorb the add (the other is ready by
        # then); it must be a plain fp16 convolution whose only consumer is this node.
        self.fused_conv = None
        if self.act:
            ex = self.ex
            steps = ex.steps
            order = lambda v: steps.index(v.producer) if v.producer in steps else -1
            me, other = (a, b) if order(a) > order(b) else (b, a)
            st = me.producer
            cons = ex.consumers.get((id(st.node), 0), []) if st is not None else []
            if (type(st).__name__ == 'ConvolutionStep' and not st.depthwise and not st.is_stem and not st.out_f32 and
                    len(cons) == 1 and getattr(st, 'fused_residual', None) is None and me is not other and
                    os.environ.get('SNIPER_FUSE_RESIDUAL', '1') != '0'):
                st.fused_residual, st.fused_dst = other, self.y
                self.fused_conv = st

    def forward(self):
        ex = self.ex
        if self.fused_conv is not None:
            return                   # written by the convolution's epilogue
        if self.act:
            n, h, w, c = self.y.nhwc()
            hip.call('sn_ew_f16', self.lhs.t, self.rhs.t, None, self.y.t, n * h * w, c, c, c, c, c, 1, hip.stream())
        else:
            hip.call('sn_ew_f32', ex.as_f32(self.lhs), ex.as_f32(self.rhs), self.y.t, self.y.t.numel(), self.code, 0.0, hip.stream())

    def backward(self):
        ex = self.ex
        g = self.y.grad
        if g is None:
            return
        fmt = self.y.fmt
        op = self.node.op
        if op in ('_plus', 'elemwise_add'):
            ex.add_grad(self.lhs, g, fmt)
            ex.add_grad(self.rhs, g, fmt)
            if self.lhs.grad is not None and self.lhs.grad is self.rhs.grad:      # one tensor, several owners: see Executor.grad_slot
                # the owners of ONE gradient tensor share one list.  Nested adds (d = a + e, a = b + c) hand d's tensor to e, b
                # and c: the inner add extends the list its own output already belongs to (minus that output: consumed here)
                group = getattr(self.y, 'grad_group', None)
                if group is None or self.y.grad is not self.lhs.grad:
                    group = []
                elif self.y in group:
                    group.remove(self.y)
                for v in (self.lhs, self.rhs):
                    if not any(v is m for m in group):
                        group.append(v)
                    v.grad_group = group
        elif op == '_minus':
            ex.add_grad(self.lhs, g, fmt)
            if self.rhs.needs_grad:
                neg = ex.empty(g.shape, F32)
                hip.call('sn_ew_f32', g, None, neg, g.numel(), 3, -1.0, hip.stream())
                ex.add_grad(self.rhs, neg, fmt)
        else:
            for me, other in ((self.lhs, self.rhs), (self.rhs, self.lhs)):
                if me.needs_grad:
                    t = ex.empty(g.shape, F32)
                    hip.call('sn_ew_f32', g, ex.as_f32(other), t, g.numel(), 2, 0.0, hip.stream())
                    ex.add_grad(me, t, fmt)
        self.y.grad = None


@register('_mul_scalar', '_plus_scalar', '_minus_scalar')
class ScalarStep(Step):
    def setup(self):
        self.x = self.ins[0]
        self.y = self.new_out('f32')
        self.y.needs_grad = self.x.needs_grad
        self.scalar = float(self.a['scalar'])

    def forward(self):
        x = self.ex.as_f32(self.x)
        if self.node.op == '_mul_scalar':
            hip.call('sn_ew_f32', x, None, self.y.t, x.numel(), 3, self.scalar, hip.stream())
        else:
            c = self.ex.const(('fill', x.numel(), self.scalar), lambda: torch.full((x.numel(),), self.scalar, device=x.device))
            hip.call('sn_ew_f32', x, c, self.y.t, x.numel(), 1 if self.node.op == '_plus_scalar' else 0, 0.0, hip.stream())

    def backward(self):
        g = self.y.grad
        if g is None or not self.x.needs_grad:
            return
        if self.node.op == '_mul_scalar':
            t = self.ex.empty(g.shape, F32)
            hip.call('sn_ew_f32', g, None, t, g.numel(), 3, self.scalar, hip.stream())
            g = t
        self.ex.add_grad(self.x, g, 'f32')
        self.y.grad = None


# ---------------------------------------------------------------------------------------------
# losses
# ---------------------------------------------------------------------------------------------
@register('SoftmaxOutput')
class SoftmaxOutputStep(Step):
    """SoftmaxOutput(multi_output, use_ignore, ignore_label, normalization, grad_scale): forward is
    the softmax; backward injects grad_scale/valid * (p - onehot) (call sites :279-281, 310-315)."""

    def setup(self):
        a = self.a
        self.x, self.label = self.ins[0], self.ins[1]
        self.multi = _bool(a.get('multi_output', False))
        self.use_ignore = _bool(a.get('use_ignore', False))
        self.ignore = float(a.get('ignore_label', -1))
        self.grad_scale = float(a.get('grad_scale', 1.0))
        self.norm = a.get('normalization', 'null')
        xs = self.x.shape
        if self.multi:
            self.outer, self.K, self.inner = xs[0], xs[1], int(np.prod(xs[2:]))
        else:
            self.outer, self.K, self.inner = int(np.prod(xs[:-1])), xs[-1], 1
        self.y = self.new_out('f32')
        self.cnt = self.ex.zeros((4,), torch.int32)

    def forward(self):
        hip.call('sn_softmax_fwd', self.ex.as_f32(self.x), self.y.t, self.outer, self.K, self.inner, hip.stream())

    def backward(self):
        ex = self.ex
        if not self.x.needs_grad:
            return
        g = ex.empty(self.x.shape, F32)
        scale = self.grad_scale
        if self.norm == 'batch':
            scale = scale / float(self.outer)
        hip.call('sn_softmax_output_bwd', self.y.t, ex.as_f32(self.label), g, self.outer, self.K, self.inner, self.ignore,
                 1 if self.use_ignore else 0, scale, 1 if self.norm == 'valid' else 0, self.cnt, hip.stream())
        ex.add_grad(self.x, g, 'f32')


@register('SoftmaxActivation')
class SoftmaxActivationStep(Step):
    def setup(self):
        self.x = self.ins[0]
        xs = self.x.shape
        if self.a.get('mode', 'instance') == 'channel':
            self.outer, self.K, self.inner = xs[0], xs[1], int(np.prod(xs[2:]))
        else:
            self.outer, self.K, self.inner = int(np.prod(xs[:-1])), xs[-1], 1
        self.y = self.new_out('f32')

    def forward(self):
        hip.call('sn_softmax_fwd', self.ex.as_f32(self.x), self.y.t, self.outer, self.K, self.inner, hip.stream())

    def backward(self):
        if self.y.grad is not None and self.x.needs_grad:
            raise NotImplementedError('SoftmaxActivation backward (inference-only in the reference graphs)')


@register('smooth_l1')
class SmoothL1Step(Step):
    def setup(self):
        self.x = self.ins[0]
        self.sigma = float(self.a.get('scalar', 1.0))
        self.y = self.new_out('f32')
        self.y.needs_grad = self.x.needs_grad

    def _consts(self, n, dev):
        z = self.ex.const(('zeros', n), lambda: torch.zeros((n,), device=dev))
        o = self.ex.const(('ones', n), lambda: torch.ones((n,), device=dev))
        return z, o

    def forward(self):
        x = self.ex.as_f32(self.x)
        z, o = self._consts(x.numel(), x.device)
        hip.call('sn_smooth_l1_loss', x, z, o, self.y.t, None, x.numel(), self.sigma, 1.0, hip.stream())

    def backward(self):
        g = self.y.grad
        if g is None or not self.x.needs_grad:
            return
        x = self.ex.as_f32(self.x)
        z, _ = self._consts(x.numel(), x.device)
        dx = self.ex.empty(x.shape, F32)
        # dpred = grad_scale * weight * f'(x) with weight := incoming gradient
        hip.call('sn_smooth_l1_loss', x, z, g, None, dx, x.numel(), self.sigma, 1.0, hip.stream())
        self.ex.add_grad(self.x, dx, 'f32')
        self.y.grad = None


@register('MakeLoss')
class MakeLossStep(Step):
    def setup(self):
        self.x = self.ins[0]
        self.grad_scale = float(self.a.get('grad_scale', 1.0))
        self.y = self.new_out('f32', alloc=False)

    def forward(self):
        self.y.t = self.ex.as_f32(self.x)

    def backward(self):
        if not self.x.needs_grad:
            return
        g = self.ex.empty(self.x.shape, F32)
        hip.call('sn_ew_f32', None, None, g, g.numel(), 4, self.grad_scale, hip.stream())
        self.ex.add_grad(self.x, g, 'f32')


# ---------------------------------------------------------------------------------------------
# proposal / RoI operators
# ---------------------------------------------------------------------------------------------
def _anchor_attrs(a):
    scales = a.get('scales', (2, 4, 7, 10, 13, 16, 24))
    ratios = a.get('ratios', (0.5, 1, 2))
    if isinstance(scales, str):
        scales = tuple(float(t) for t in scales.strip('()[] ').split(',') if t.strip())
    if isinstance(ratios, str):
        ratios = tuple(float(t) for t in ratios.strip('()[] ').split(',') if t.strip())
    return tuple(scales), tuple(ratios)


class _ProposalBase(Step):
    def common(self):
        from ..data.anchors import generate_anchors
        ex, a = self.ex, self.a
        self.cls, self.bbox, self.info = self.data_in('cls_prob'), self.data_in('bbox_pred'), self.data_in('im_info')
        # (B, 2, A*Fh, Fw) in the training graph, (B, 2A, Fh, Fw) at test time: same memory; the feature map's own
        # geometry comes from bbox_pred (B, 4A, Fh, Fw) -- test images are not square
        B = self.cls.shape[0]
        _, c4a, Fh, Fw = self.bbox.shape
        self.B, self.F, self.Fh, self.Fw = B, Fw, Fh, Fw
        self.A = c4a // 4
        assert int(np.prod(self.cls.shape[1:])) == 2 * self.A * Fh * Fw, (self.cls.shape, self.bbox.shape)
        self.stride = int(a.get('feature_stride', 16))
        scales, ratios = _anchor_attrs(a)
        if len(scales) * len(ratios) != self.A:
            raise ValueError('%s: %d anchors per cell in cls_prob but %d scales x %d ratios' % (
                self.node.name, self.A, len(scales), len(ratios)))
        base = generate_anchors(self.stride, list(ratios), list(np.array(scales, np.float32))).astype(np.float32)
        self.base = torch.from_numpy(base).to(ex.device)
        self.pre = int(a.get('rpn_pre_nms_top_n', 6000))
        self.post = int(a.get('rpn_post_nms_top_n', 300))
        self.thresh = float(a.get('threshold', 0.7))
        self.min_size = float(a.get('rpn_min_size', 0))
        nbytes = hip.query('sn_proposal_workspace_bytes', B, self.A, Fh, Fw, self.pre, self.post)
        self.wsbuf = ex.empty((nbytes,), torch.uint8)


@register('MultiProposal')
class MultiProposalStep(_ProposalBase):
    def setup(self):
        self.common()
        self.rois = self.new_out('f32', 0)
        self.scores = self.new_out('f32', 1)

    def forward(self):
        ex = self.ex
        hip.call('sn_multi_proposal', ex.as_f32(self.cls), ex.as_f32(self.bbox), ex.as_f32(self.info), self.base, self.B, self.A,
                 self.Fh, self.Fw, self.stride, self.pre, self.post, self.thresh, self.min_size, self.wsbuf, self.rois.t, self.scores.t,
                 hip.stream())


@register('MultiProposalTarget')
class MultiProposalTargetStep(_ProposalBase):
    def setup(self):
        self.common()
        self.gt, self.vr = self.data_in('gt_boxes'), self.data_in('valid_ranges')
        self.G = self.gt.shape[1]
        self.fg = float(self.a.get('fg_thresh', 0.5))
        self.stds = np.array(self.a.get('bbox_stds', (0.1, 0.1, 0.2, 0.2)), np.float32)
        self.outs = [self.new_out('f32', i) for i in range(4)]

    def forward(self):
        ex = self.ex
        rois, label, tgt, wgt = [o.t for o in self.outs]
        hip.call('sn_multi_proposal_target', ex.as_f32(self.cls), ex.as_f32(self.bbox), ex.as_f32(self.info), ex.as_f32(self.gt),
                 ex.as_f32(self.vr), self.base, self.B, self.A, self.Fh, self.Fw, self.stride, self.G, self.pre, self.post, self.thresh,
                 self.min_size, self.fg, self.stds.ctypes.data, self.wsbuf, rois, label, tgt, wgt, hip.stream())


@register('MultiProposalTargetMask')
class MultiProposalTargetMaskStep(_ProposalBase):
    """resnet_mx_101_e2e_mask.py:317-318: MultiProposalTarget + the mask RoIs of every chip and the GT row each matched."""

    def setup(self):
        self.common()
        self.gt, self.vr = self.data_in('gt_boxes'), self.data_in('valid_ranges')
        self.G = self.gt.shape[1]
        self.fg = float(self.a.get('fg_thresh', 0.5))
        self.stds = np.array(self.a.get('bbox_stds', (0.1, 0.1, 0.2, 0.2)), np.float32)
        self.outs = [self.new_out('f32', i) for i in range(6)]
        self.nm = self.outs[4].shape[0] // self.B
        self.match = self.ex.empty((self.B * self.post,), F32)

    def forward(self):
        ex = self.ex
        rois, label, tgt, wgt, mrois, mids = [o.t for o in self.outs]
        hip.call('sn_multi_proposal_target_mask', ex.as_f32(self.cls), ex.as_f32(self.bbox), ex.as_f32(self.info), ex.as_f32(self.gt),
                 ex.as_f32(self.vr), self.base, self.B, self.A, self.Fh, self.Fw, self.stride, self.G, self.pre, self.post, self.thresh,
                 self.min_size, self.fg, self.stds.ctypes.data, self.wsbuf, self.match, self.nm, rois, label, tgt, wgt, mrois, mids,
                 hip.stream())


@register('MaskRcnnTarget')
class MaskRcnnTargetStep(Step):
    """:392-395: rasterise the matched GT polygon into every mask RoI's 28 x 28 grid (labels only, no gradient)."""

    def setup(self):
        a = self.a
        self.rois, self.polys, self.ids = self.data_in('rois'), self.data_in('mask_polys'), self.data_in('mask_ids')
        self.ms = int(a.get('mask_size', 28))
        self.B, self.max_gts, self.max_len = self.polys.shape
        self.N = self.rois.shape[0]
        if self.N % self.B:
            raise ValueError('%s: %d mask RoIs for %d chips' % (self.node.name, self.N, self.B))
        self.outs = [self.new_out('f32', 0), self.new_out('f32', 1)]

    def forward(self):
        ex = self.ex
        hip.call('sn_mask_rcnn_target', ex.as_f32(self.rois), ex.as_f32(self.polys), ex.as_f32(self.ids), self.N, self.N // self.B,
                 self.max_gts, self.max_len, self.ms, self.outs[0].t, self.outs[1].t, hip.stream())


@register('Deconvolution')
class DeconvolutionStep(Step):
    """2x2 / stride-2 up-sampling of the mask head (:247-248): every input pixel writes its own 2x2 output block, so the
    operator is a 1x1 convolution to 4*Cout channels (weight rows ordered (a, b, o), the implicit-GEMM kernels) followed by
    a depth-to-space shuffle."""

    def setup(self):
        ex, a = self.ex, self.a
        self.x = self.data_in()
        k, s, p = _tup(a['kernel']), _tup(a.get('stride', (1, 1))), _tup(a.get('pad', (0, 0)))
        if k != (2, 2) or s != (2, 2) or p != (0, 0) or int(a.get('num_group', 1)) != 1 or 'bias' in self.slots:
            raise NotImplementedError('%s: Deconvolution other than kernel 2x2 / stride 2 / no pad / no bias' % self.node.name)
        self.N, self.C, self.H, self.W = self.x.shape
        self.O = int(a['num_filter'])
        self.w = ex.register_param(self.pname('weight'), 'deconv', None, need_wT=ex.for_training and self.x.needs_grad)
        self.y = self.new_out('act')
        self.y.needs_grad = ex.for_training and (self.x.needs_grad or self.w.trainable)
        self.tmp = ex.empty((self.N, self.H, self.W, 4 * self.O), F16)

    def forward(self):
        ex = self.ex
        O4 = 4 * self.O
        hip.call('sn_conv_fwd', ex.as_act(self.x), self.w.w16, None, None, self.tmp, self.N, self.H, self.W, self.C, self.C, O4, O4, 0,
                 1, 1, 1, 0, 1, 0, 0, hip.stream())
        hip.call('sn_depth_to_space2', self.tmp, self.y.t, self.N, self.H, self.W, self.O, 0, hip.stream())

    def backward(self):
        ex = self.ex
        if self.y.grad is None or not self.y.needs_grad:
            return
        g = self.y.grad
        O4 = 4 * self.O
        dtmp = ex.empty((self.N, self.H, self.W, O4), F16)
        hip.call('sn_space_to_depth2', g, dtmp, self.N, self.H, self.W, self.O, hip.stream())
        x = ex.as_act(self.x)
        if self.w.trainable:
            ex.on_side(lambda: _wgrad(ex, dtmp, x, self.w.grad, self.N, self.H, self.W, self.C, self.C, O4, O4, 1, 1, 1, 0, 1),
                       keep=(dtmp, x))
        if self.x.needs_grad:
            dx, acc = ex.grad_slot(self.x) if self.x.fmt == 'act' else (ex.empty((self.N, self.H, self.W, self.C), F16), None)
            hip.call('sn_conv_dgrad', dtmp, self.w.wT16, acc, dx, self.N, self.H, self.W, self.C, self.C, O4, O4, self.C,
                     1, 1, 1, 0, 1, 0, hip.stream())
            if self.x.fmt != 'act':
                ex.add_grad(self.x, dx, 'act')
        self.y.grad = None


@register('pick')
class PickStep(Step):
    """mx.sym.pick(data, index, axis=1, keepdims=True) (:398-399): the RoI class's mask map."""

    def setup(self):
        a = self.a
        self.x, self.idx = self.data_in('data'), self.data_in('index')
        if int(a.get('axis', -1)) != 1 or not _bool(a.get('keepdims', False)) or len(self.x.shape) != 4 or self.x.fmt != 'act':
            raise NotImplementedError('%s: pick other than axis=1, keepdims=True on an activation tensor' % self.node.name)
        self.y = self.new_out('act')
        self.y.needs_grad = self.x.needs_grad

    def forward(self):
        n, h, w, c = self.x.nhwc()
        hip.call('sn_pick_fwd', self.x.t, self.ex.as_f32(self.idx), self.y.t, n, h * w, c, hip.stream())

    def backward(self):
        if self.y.grad is None or not self.x.needs_grad:
            return
        n, h, w, c = self.x.nhwc()
        dx, acc = self.ex.grad_slot(self.x)
        if acc is not None and acc is not dx:        # the scatter adds in place: bring the earlier contributions over first
            hip.call('sn_copy2d', acc, dx, 1, acc.numel(), acc.numel(), acc.numel(), 0, 0, hip.stream())
        hip.call('sn_pick_bwd', self.y.grad, self.ex.as_f32(self.idx), dx, n, h * w, c, 1 if acc is not None else 0, hip.stream())
        self.y.grad = None


@register('DeformablePSROIPooling')
class DPSROIPoolStep(Step):
    """group_size 1 (the SNIPER heads, :286-293): sn_dpsroi_pool_*; group_size G > 1 (position-sensitive R-FCN head,
    BASELINE config C4): sn_psroi_pool_*, data channels (output_dim, G, G)."""

    def setup(self):
        ex, a = self.ex, self.a
        self.x, self.rois = self.data_in('data'), self.data_in('rois')
        self.no_trans = _bool(a.get('no_trans', False))
        self.trans = None if self.no_trans or 'trans' not in self.slots else self.data_in('trans')
        self.P, self.S = int(a['pooled_size']), int(a.get('sample_per_part', 1))
        self.G, self.D = int(a.get('group_size', 1)), int(a['output_dim'])
        if int(a.get('part_size', 0) or self.P) != self.P:
            raise NotImplementedError('%s: DeformablePSROIPooling with part_size != pooled_size' % self.node.name)
        if self.D * self.G * self.G != self.x.shape[1]:
            raise ValueError('%s: data must carry output_dim * group_size^2 = %d channels (got %d)' %
                             (self.node.name, self.D * self.G * self.G, self.x.shape[1]))
        if self.trans is not None and tuple(self.trans.shape[1:]) != (2, self.P, self.P):
            raise NotImplementedError('%s: per-class offset fields (trans %s); class-agnostic (R,2,P,P) only' %
                                      (self.node.name, (self.trans.shape,)))
        self.scale = float(a['spatial_scale'])
        self.tstd = float(a.get('trans_std', 0.0))
        self.y = self.new_out('act')
        self.y.needs_grad = self.x.needs_grad or (self.trans is not None and self.trans.needs_grad)
        self.ws = None
        # Position-sensitive maps GROUP-MAJOR (round 6): in the operator's channel order (d, gh, gw) the output_dim channels of one bin
        # lie group_size^2 apart, and every bin drags the whole pixel through L2 (sn_psroi_pool_fwd: 7.5 ms per call on the 7*7*81 map
        # at 16 x 300 RoIs).  When the map comes straight out of a convolution that nobody else reads, that convolution writes its
        # output channels in (gh, gw, d) order instead -- a permutation of its weight ROWS and bias in the kernels' layout
        # (Param.out_perm; reference order at the checkpoint boundary as ever) -- and the pooling kernels index (gh*G + gw)*D + d.
        self.gm = 0
        prod = self.x.producer
        if (self.G > 1 and os.environ.get('SNIPER_PS_GROUP_MAJOR', '1') != '0' and type(prod).__name__ == 'ConvolutionStep'
                and self.x.fmt == 'act' and not prod.depthwise and not prod.out_f32 and getattr(prod, 'fold_bn', None) is None
                and len(ex.consumers.get((id(prod.node), 0), [])) == 1 and (id(prod.node), 0) not in ex.head_keys):
            G, D = self.G, self.D
            perm = np.empty(D * G * G, np.int64)
            for g in range(G * G):
                for d in range(D):
                    perm[g * D + d] = d * G * G + g
            for q in (prod.w, prod.b):
                if q is not None:
                    q.out_perm = perm
            self.x.chan_perm = perm
            self.gm = 1

    def forward(self):
        ex = self.ex
        n, h, w, c = self.x.nhwc()
        R = self.rois.shape[0]
        trans = ex.as_f32(self.trans) if self.trans else None
        if self.G == 1:
            hip.call('sn_dpsroi_pool_fwd_images', ex.as_act(self.x), ex.as_f32(self.rois), trans, self.y.t, R, n, h, w, c, self.P,
                     self.S, self.scale, self.tstd, hip.stream())
        else:
            hip.call('sn_psroi_pool_fwd', ex.as_act(self.x), ex.as_f32(self.rois), trans, self.y.t, R, h, w, self.D, self.G,
                     self.P, self.S, self.scale, self.tstd, self.gm, hip.stream())

    def backward(self):
        ex = self.ex
        if self.y.grad is None:
            return
        n, h, w, c = self.x.nhwc()
        R = self.rois.shape[0]
        d16 = ex.empty((n, h, w, c), F16)
        d_trans = ex.empty(self.trans.shape, F32) if self.trans is not None else None
        if self.ws is None:
            self.ws = ex.empty((hip.query('sn_dpsroi_bwd_workspace_bytes', R),), torch.uint8)
        trans = ex.as_f32(self.trans) if self.trans else None
        if self.G == 1:
            hip.call('sn_dpsroi_pool_bwd', self.y.grad, ex.as_act(self.x), ex.as_f32(self.rois), trans, d16, 0, d_trans, R, n, h,
                     w, c, self.P, self.S, self.scale, self.tstd, self.ws, hip.stream())
        else:
            hip.call('sn_psroi_pool_bwd', self.y.grad, ex.as_act(self.x), ex.as_f32(self.rois), trans, d16, 0, d_trans, R, n, h,
                     w, self.D, self.G, self.P, self.S, self.scale, self.tstd, self.gm, self.ws, hip.stream())
        if self.x.needs_grad:
            ex.add_grad(self.x, d16, 'act')
        if self.trans is not None and self.trans.needs_grad:
            ex.add_grad(self.trans, d_trans, 'f32')
        self.y.grad = None


@register('Custom')
class CustomStep(Step):
    """mx.operator.CustomOp plugins run on the host (numpy NDArrays), like MXNet's python ops."""

    def setup(self):
        from ..mx import operator as _operator
        self.prop = _operator.get_prop(self.a.get('op_type'), self.a)
        shapes = [v.shape for v in self.ins]
        self.op = self.prop.create_operator(None, shapes, None)
        self.outs = [self.new_out('f32', i) for i in range(self.node.num_outputs)]

    def forward(self):
        from ..mx import ndarray as nd
        ins = [nd.NDArray(self.ex.as_f32(v).cpu().numpy()) for v in self.ins]
        outs = [nd.zeros(o.shape) for o in self.outs]
        self.op.forward(self.ex.is_train, ['write'] * len(outs), ins, outs, [])
        for o, h in zip(self.outs, outs):
            o.t.copy_(torch.from_numpy(h.asnumpy()))
        self._host = (ins, outs)

    def backward(self):
        """CustomOp.backward on the host (box_annotator_ohem.py:80-83 assigns zeros): out_grad carries the gradients that
        reached the outputs (zeros where none did -- `need_top_grad=False` plugins ignore them anyway), every input that
        wants a gradient receives the plugin's in_grad."""
        from ..mx import ndarray as nd
        if not any(v.needs_grad for v in self.ins):
            for o in self.outs:
                o.grad = None
            return
        ins, outs = self._host
        out_grad = [nd.NDArray(o.grad.float().cpu().numpy().reshape(o.shape)) if o.grad is not None else nd.zeros(o.shape) for o in self.outs]
        in_grad = [nd.zeros(v.shape) for v in self.ins]
        self.op.backward(['write'] * len(in_grad), out_grad, ins, outs, in_grad, [])
        for v, g in zip(self.ins, in_grad):
            if v.needs_grad:
                self.ex.add_grad(v, torch.from_numpy(np.ascontiguousarray(g.asnumpy(), np.float32)).to(self.ex.device), 'f32')
        for o in self.outs:
            o.grad = None
