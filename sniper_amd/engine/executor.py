"""Graph executor: lowers a captured symbol graph (sniper_amd.mx.symbol) onto the HIP kernels of
libsniper_hip.so and runs forward / backward / SGD on one GPU.

This is the replacement for what ``mx.mod.Module`` delegates to in the un-vendored SNIPER-mxnet
runtime (component 10): executor, operator kernels, optimizer.  Design, MI355X first:

* activations are channels-last fp16 ("act", physical (N,H,W,C)), because the MFMA implicit-GEMM
  wants the contraction dimension contiguous; loss-side tensors are fp32 in the reference's NCHW
  order ("f32").  Conversions are inserted only where the graph crosses between the two worlds
  (RPN heads -> Reshape/softmax/proposal ops, FC logits -> losses).
* parameters live in flat fp32 arenas (master weights, gradients, momentum) in the kernels' layout
  [Cout][KH*KW][Cin]; one memset zeroes all gradients, one RCCL all-reduce sums them, a handful of
  fused SGD launches update them and emit the fp16 compute copies.
* torch provides device memory, streams and (elsewhere) torch.distributed; every byte of compute
  goes through the C ABI (sniper_amd.hip.call).  No CPU fallback.
"""
import os
import warnings

import numpy as np
import torch

from .. import hip
from .shapes import infer_shapes

F16, F32 = torch.float16, torch.float32


def _pad8(n):
    return (n + 7) // 8 * 8


class Val(object):
    """One tensor of the graph.  fmt 'act': fp16 channels-last, t has shape (N,H,W,C) (2-D logical
    tensors use H=W=1); fmt 'f32': fp32, reference order, t has the logical shape."""
    __slots__ = ('name', 'shape', 'fmt', 't', 'needs_grad', 'grad', 'alt', 'stem', 'consumers', 'producer', 'grad_group', 'chan_perm')

    def __init__(self, name, shape, fmt):
        self.name, self.shape, self.fmt = name, tuple(shape), fmt
        self.t = None
        self.chan_perm = None    # device channel j holds reference channel chan_perm[j] (a group-major position-sensitive map), or None
        self.needs_grad = False
        self.grad = None
        self.alt = None      # cached other-format copy (forward)
        self.stem = None     # (src f32 NCHW Val, scale, shift) for a BN-folded image input
        self.consumers = 0
        self.producer = None  # the Step whose output this is
        self.grad_group = None   # the Vals that were handed the SAME gradient tensor by residual adds (one shared list; Executor.grad_slot)

    def nhwc(self):
        s = self.shape
        return (s[0], s[2], s[3], s[1]) if len(s) == 4 else (s[0], 1, 1, int(np.prod(s[1:])))


class Param(object):
    __slots__ = ('name', 'ref_shape', 'kind', 'int_shape', 'trainable', 'lr_mult', 'wd_mult', 'master', 'grad', 'mom',
                 'w16', 'wT16', 'need_wT', 'offset', 'numel', 'fc_in', 'half_region', 'step_index', 'phase', 'out_perm')

    def __init__(self, name, ref_shape):
        self.name, self.ref_shape = name, tuple(ref_shape)
        self.kind, self.int_shape = 'vec', tuple(ref_shape)
        self.trainable, self.lr_mult, self.wd_mult = True, 1.0, 1.0
        self.master = self.grad = self.mom = self.w16 = self.wT16 = None
        self.need_wT = False
        self.fc_in = None
        self.half_region = False   # weight of an operator the reference graph runs in fp16 (between its Cast nodes)
        self.step_index = 0        # forward index of the (first) step that owns it
        self.phase = 0             # 0: its gradient is complete after the first backward segment, 1: after the second
        self.numel = int(np.prod(ref_shape))
        # permutation of the OUTPUT channels (axis 0), internal row i = reference row out_perm[i]: a convolution whose output map a
        # position-sensitive RoI pooling reads writes it group-major (ops.py DPSROIPoolStep); None = reference order
        self.out_perm = None

    # reference layout <-> kernel layout
    def to_internal(self, a):
        a = np.asarray(a, np.float32).reshape(self.ref_shape)
        if self.out_perm is not None:
            a = a[self.out_perm]
        if self.kind == 'conv':
            return np.ascontiguousarray(a.transpose(0, 2, 3, 1))
        if self.kind == 'deconv':    # (Cin, Cout, 2, 2) -> rows (a, b, o) of a 1x1 convolution: [4*Cout][1][Cin]
            return np.ascontiguousarray(a.transpose(2, 3, 1, 0)).reshape(self.int_shape)
        if self.kind == 'stem':      # (O, C<=4, KH, KW) -> [O][KH][KWP*4], zero padded
            kh, kw, kwp = self.fc_in
            out = np.zeros((a.shape[0], kh, kwp, 4), np.float32)
            out[:, :, :kw, :a.shape[1]] = a.transpose(0, 2, 3, 1)
            return out.reshape(self.int_shape)
        if self.kind == 'fc' and self.fc_in is not None:
            c, h, w = self.fc_in
            return np.ascontiguousarray(a.reshape(a.shape[0], c, h, w).transpose(0, 2, 3, 1))
        return a

    def to_reference(self, a):
        if self.out_perm is not None:
            perm, self.out_perm = self.out_perm, None
            try:
                rows = self.to_reference(a)          # reference layout, rows still in internal order
            finally:
                self.out_perm = perm
            out = np.empty_like(rows)
            out[perm] = rows
            return out
        a = np.asarray(a, np.float32)
        if self.kind == 'conv':
            o, i, kh, kw = self.ref_shape
            return np.ascontiguousarray(a.reshape(o, kh, kw, i).transpose(0, 3, 1, 2))
        if self.kind == 'deconv':
            ci, co, kh, kw = self.ref_shape
            return np.ascontiguousarray(a.reshape(kh, kw, co, ci).transpose(3, 2, 0, 1))
        if self.kind == 'stem':
            o, i, kh, kw = self.ref_shape
            return np.ascontiguousarray(a.reshape(o, kh, self.fc_in[2], 4)[:, :, :kw, :i].transpose(0, 3, 1, 2))
        if self.kind == 'fc' and self.fc_in is not None:
            c, h, w = self.fc_in
            return np.ascontiguousarray(a.reshape(-1, h, w, c).transpose(0, 3, 1, 2)).reshape(self.ref_shape)
        return a.reshape(self.ref_shape)


_F32_CONSUMERS = {'Reshape', 'SoftmaxOutput', 'SoftmaxActivation', 'smooth_l1', 'MakeLoss', 'MultiProposal',
                  'MultiProposalTarget', 'BlockGrad', '_mul_scalar', '_plus_scalar', '_minus_scalar', 'Flatten', 'Custom'}
_PRODUCES_ACT = {'Convolution', 'FullyConnected', 'BatchNorm', 'Activation', 'Pooling', 'Concat', 'DeformableConvolution',
                 'DeformablePSROIPooling', 'clip', 'Deconvolution', 'pick'}


_CAPTURE_ALLOWED = True


def set_capture_allowed(flag):
    """Process-wide switch for NEW hipGraph captures (replays of captured graphs are unaffected).  Off while several threads drive
    the GPU at once (the concurrent test-time jobs of sniper_amd.inference): a capture is not safe beside another thread's
    allocations / synchronisations ("operation failed due to a previous error during capture"), so an executor that has no
    graph yet runs eagerly until the jobs are back to one at a time."""
    global _CAPTURE_ALLOWED
    _CAPTURE_ALLOWED = bool(flag)


def settle_heap():
    """A bound executor is some 10^5 long-lived Python objects (steps, values, parameters, plans) that reference each other.  The
    cyclic collector walks every tracked object at each full collection: 4 ms per bound executor, 230 ms per collection with the
    61 test-time executors of a 160-image pass, one collection per pass -- and 190 ms before EVERY forward capture
    (profiles/r04_infer_long_pass.txt: 7.7 s of a first pass).  After a build / a capture the heap is collected once and moved to the
    permanent generation (gc.freeze): reference counting still frees whatever is dropped, the collector no longer walks what
    will not die.  Module._evict_stale thaws before it drops executors (their cycles need the collector).  SNIPER_GC_FREEZE=0:
    leave the collector alone."""
    if os.environ.get('SNIPER_GC_FREEZE', '1') == '0':
        return
    import gc
    gc.collect()
    gc.freeze()
    _SETTLED[0] += 1


_SETTLED = [0]          # settle_heap calls since the last resettle_heap


def resettle_heap():
    """End of a pass (sniper_amd.inference.imdb_detection_wrapper, after the pass's own objects are dropped): settle_heap runs
    INSIDE a pass -- Module.forward builds / captures when a batch has a new shape -- so whatever is alive at that moment is frozen
    with the executors: the pass's iterators, its image cache (up to SNIPER_IMAGE_CACHE_GB of device images), its batches.  What of
    that has died by the end of the pass and sits in a reference cycle would stay until the next thaw; so a pass that settled
    anything thaws, collects and freezes again once, when only the long-lived objects are left.  A pass over shapes already bound
    does nothing here."""
    if _SETTLED[0] == 0 or os.environ.get('SNIPER_GC_FREEZE', '1') == '0':
        return
    import gc
    _SETTLED[0] = 0
    gc.unfreeze()
    gc.collect()
    gc.freeze()


def thaw_heap():
    """Before a Module binds: objects frozen by settle_heap that have since been dropped (a Module rebuilt per call, as the
    reference's detect_scale_worker does -- its executors, pool and parameters are cycles) are only reclaimed by a collection that
    can see them."""
    import gc
    if gc.get_freeze_count() > 0:
        gc.unfreeze()
        gc.collect()


class ActivationPool(object):
    """Forward activations of the TEST-TIME executors of one Module, in memory they share.  A Module runs one executor at a time
    (one per bucketed batch shape, all on the Module's stream) and a forward writes every activation before it reads it, so the
    executors of every shape can lay their activations over the same bytes: what a Module holds is the footprint of its largest
    shape, not the sum over the shapes it has seen (a 2 x 1408 x 2048 R101 forward is 9 GB of activations; 33 bound shapes held
    184 GB of a 288 GB card and the cache evicted on every batch -- profiles/r04_infer_long_pass.txt).  MXNet's `reshape`
    shares the memory of the bound executor the same way.  The pool is a list of buffers that never move (captured forwards
    hold addresses inside them); every executor walks the list from the start with its own cursor, and a request that does not
    fit the remaining buffers appends one."""
    CHUNK = 1 << 30
    ALIGN = 256

    def __init__(self, device):
        self.device = device
        self.buffers = []

    def cursor(self):
        return [0, 0]          # buffer index, byte offset

    def take(self, cur, shape, dtype):
        shape = tuple(int(x) for x in shape)
        n = 1
        for x in shape:
            n *= x
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        need = max(self.ALIGN, -(-nbytes // self.ALIGN) * self.ALIGN)
        while True:
            if cur[0] == len(self.buffers):
                self.buffers.append(torch.empty((max(self.CHUNK, need) + self.ALIGN,), dtype=torch.uint8, device=self.device))
            buf = self.buffers[cur[0]]
            if cur[1] == 0:
                cur[1] = (-buf.data_ptr()) % self.ALIGN        # addresses, not offsets, are aligned
            if cur[1] + need <= buf.numel():
                t = buf[cur[1]:cur[1] + nbytes].view(dtype).view(shape)
                cur[1] += need
                return t
            cur[0], cur[1] = cur[0] + 1, 0

    def nbytes(self):
        return sum(b.numel() for b in self.buffers)


class Executor(object):
    def __init__(self, symbol, input_shapes, for_training=True, fixed_param_names=(), device=None, data_names=None,
                 label_names=None, loss_scale_hint=None, split_backward=False, act_pool=None, share_params=None):
        self.sym = symbol
        self.device = device or hip.require_gpu()
        # test-time only: the parameters (fp32 masters, fp16 copies, moving statistics, BatchNorm-folded weights) of another bound
        # shape of the same Module are THIS executor's too -- one set per Module, not one per batch shape (0.5 GB and 70 ms of
        # packing per shape for R101).  MXNet's `reshape` binds the new shape with shared parameter arrays the same way.
        self._share = share_params if (share_params is not None and not for_training and not share_params.for_training) else None
        self.fold_store = self._share.fold_store if self._share is not None else {}      # weight name -> (folded w16, folded bias)
        self.shared_names = set()
        # test-time only: forward activations carved out of the Module's shared pool (ActivationPool)
        self.act_pool = None if for_training else act_pool
        self._act_cursor = self.act_pool.cursor() if self.act_pool is not None else None
        self.for_training = for_training
        self.nodes = symbol._topo()
        self.shapes = infer_shapes(symbol, input_shapes)
        self.input_names = list(input_shapes.keys())
        self.fixed = set(fixed_param_names or ())
        self.arg_names = symbol.list_arguments()
        self.aux_names = symbol.list_auxiliary_states()
        self.param_names = [n for n in self.arg_names if n not in input_shapes]
        self.vals = {}        # (id(node), idx) -> Val
        self.params = {}      # name -> Param
        self.aux = {}         # name -> torch fp32 vector
        self.steps = []       # lowered ops, forward order
        self.ws = hip.Workspace()
        self._const = {}
        self._bn_cache_valid = False
        self._wt_tables = {}
        self.num_update = 0
        self._lower()
        self._mark_half_region()
        self.split_k = self._choose_split() if (split_backward and for_training) else 0
        self._alloc_params()
        # HIP graphs: a training step is ~2500 launches of static shape on preallocated buffers -- launch-bound on
        # the host (43 ms of enqueue per 47 ms step measured eagerly).  After `graph_warmup` eager steps the
        # forward+backward pass and the optimizer pass are each captured once (torch.cuda.CUDAGraph = hipGraph on
        # torch's stream, which is the stream every C-ABI call is given) and replayed; the gradient all-reduce runs
        # between the two.  Hyper-parameters that change per step live in `self.hyper` on the device.
        self.hyper = self.zeros((4,), F32)      # lr, wd, momentum, rescale_grad
        self.use_graphs = (for_training and os.environ.get('SNIPER_HIP_GRAPHS', '1') != '0' and
                           not any(type(st).__name__ == 'CustomStep' for st in self.steps))
        self.graph_warmup = 2
        # Inference executors (one per input shape, kept by the Module) replay a captured forward from their third call on:
        # an eager test-time forward is ~250 launches per batch from Python, host-bound at the finer AutoFocus scales.
        self.use_infer_graphs = (not for_training and os.environ.get('SNIPER_HIP_GRAPHS', '1') != '0' and
                                 not any(type(st).__name__ == 'CustomStep' for st in self.steps))
        self._infer_graph, self._infer_calls = None, 0
        self._keepalive = []
        # Weight gradients are DEFERRED and launched as tables of layers (sn_conv_wgrad_batch): one stage-3 layer has 16-36
        # output tiles, so alone it needs a 7-16-way K split with fp32 partial slabs; a few dozen layers per launch fill the
        # 256 CUs with whole-K jobs (csrc/conv_wgrad_ps.hip).  A queued layer keeps its dY / X tensors alive; a dY that another
        # Val would accumulate into IN PLACE before the flush (the residual trunk's shared gradient) is protected by
        # grad_slot (out-of-place accumulation).  SNIPER_WGRAD_DEFER=0: launch per layer, at the layer's backward.
        # (Rounds 1-2 ran the weight gradients on a second stream instead; measured, the overlap bought nothing -- 30.15 ms with,
        # 29.72 ms without, profiles/r02_* -- and round 3 removed it.)
        self.defer_wgrads = for_training and os.environ.get('SNIPER_WGRAD_DEFER', '1') != '0'
        self._pending_wgrads = []    # [(dy, x, dw, N, H, W, C, x_ps, O, dy_ps, kh, kw, stride, pad, dil)]
        self._pending_reads = set()  # storages queued weight gradients still have to read
        self._graph_fb = self._graph_up = None
        self._eager_fb = self._eager_up = 0

    # ------------------------------------------------------------------------------------------
    # helpers
    # ------------------------------------------------------------------------------------------
    def empty(self, shape, dtype):
        return torch.empty(tuple(int(s) for s in shape), dtype=dtype, device=self.device)

    def zeros(self, shape, dtype):
        return torch.zeros(tuple(int(s) for s in shape), dtype=dtype, device=self.device)

    def act_empty(self, shape, dtype):
        """A forward activation (a step's output, an input buffer, a column buffer): rewritten by every forward before it is
        read.  From the Module's pool at test time, a tensor of its own otherwise."""
        if self.act_pool is None:
            return self.empty(shape, dtype)
        return self.act_pool.take(self._act_cursor, shape, dtype)

    def const(self, key, fn):
        if key not in self._const:
            self._const[key] = fn()
        return self._const[key]

    def vals_shape(self, input_name):
        """bound shape of a graph input"""
        return self.shapes[('var', input_name)]

    def val_of(self, node, idx=0):
        return self.vals[(id(node), idx)]

    def in_vals(self, node):
        return [self.vals[(id(n), i)] for n, i in node.inputs]

    # ---- format conversion (forward) ----------------------------------------------------------
    def as_f32(self, v):
        """fp32 reference-order view of a Val (converted copy cached per forward)."""
        if v.fmt == 'f32':
            return v.t
        if v.alt is None:
            n, h, w, c = v.nhwc()
            out = self.empty(v.shape, F32)
            if h * w == 1:
                hip.call('sn_copy2d', v.t, out, n, c, c, c, 0, 1, hip.stream())
            else:
                hip.call('sn_transpose_batched', v.t, out, n, h * w, c, h * w * c, c * h * w, c, h * w, 0, 1, hip.stream())
            if v.chan_perm is not None:      # (a head / a test reading a group-major map: back to the reference's channel order)
                inv = np.empty(len(v.chan_perm), np.int64)
                inv[np.asarray(v.chan_perm)] = np.arange(len(v.chan_perm))
                out = out.index_select(1, torch.from_numpy(inv).to(out.device))
            v.alt = out
        return v.alt

    def as_act(self, v):
        if v.fmt == 'act':
            return v.t
        if v.alt is None:
            n, h, w, c = v.nhwc()
            out = self.empty((n, h, w, c), F16)
            if h * w == 1:
                hip.call('sn_copy2d', v.t, out, n, c, c, c, 1, 0, hip.stream())
            else:
                hip.call('sn_transpose_batched', v.t, out, n, c, h * w, c * h * w, h * w * c, h * w, c, 1, 0, hip.stream())
            v.alt = out
        return v.alt

    # ---- gradient plumbing -------------------------------------------------------------------
    def grad_slot(self, v):
        """-> (dst, src): the tensor (v's own format) the caller writes v's gradient contribution into, and the tensor it adds
        to it (None for the first writer; normally dst itself -- later writers add in place).
        A gradient tensor can be shared: a residual add hands the SAME tensor to both operands (add_grad), so the tensor
        about to be accumulated into may be the dY of a convolution whose weight gradient has not run yet -- queued for a
        batched launch: then the sum goes to a fresh tensor, src = the old one, which the queue keeps alive."""
        if v.grad is None:
            v.grad = self.empty(v.t.shape, v.t.dtype)
            return v.grad, None
        # copy on write: the tensor is also the (not yet consumed) gradient of the residual add's other operand.  In ResNet /
        # MobileNetV2 that operand's producer has run its backward and dropped the tensor by now (no copy); any other
        # graph order gets a private copy instead of a silently corrupted dY.
        group = getattr(v, 'grad_group', None)
        if group is not None:
            v.grad_group = None
            ptr = v.grad.data_ptr()
            if any(m is not v and m.grad is not None and m.grad.data_ptr() == ptr for m in group):
                v.grad = v.grad.clone()          # (v leaves the group: its tensor is its own from here on)
            group[:] = [m for m in group if m is not v]
        if self._pending_reads and v.grad.untyped_storage().data_ptr() in self._pending_reads:
            src = v.grad
            v.grad = self.empty(src.shape, src.dtype)
            return v.grad, src
        return v.grad, v.grad

    def queue_wgrad(self, *problem):
        """problem = the arguments of sn_conv_wgrad up to `dil` (tensors dy, x, dw first)."""
        # two layers sharing one weight tensor must not sit in one table: unsplit jobs do a plain `dw += tile`, so the two
        # problems' tiles would race on that gradient (the reference's symbols share no weights; a custom graph may)
        if any(q[2].data_ptr() == problem[2].data_ptr() for q in self._pending_wgrads):
            self.flush_wgrads()
        self._pending_wgrads.append(problem)
        for t in problem[:2]:
            self._pending_reads.add(t.untyped_storage().data_ptr())
        if len(self._pending_wgrads) >= 160:
            self.flush_wgrads()

    def flush_wgrads(self):
        if not self._pending_wgrads:
            return
        tab = hip.wgrad_table(self._pending_wgrads)
        n = len(self._pending_wgrads)
        need = hip.query('sn_conv_wgrad_batch_workspace_bytes', tab, n)
        ws = self.ws.get(need) if need else None
        hip.call('sn_conv_wgrad_batch', tab, n, ws, need, hip.stream())
        self._pending_wgrads = []
        self._pending_reads.clear()

    def add_grad(self, v, g, g_fmt):
        """Accumulate gradient tensor g (format g_fmt: 'act' NHWC fp16 or 'f32' reference order) into v."""
        if not v.needs_grad:
            return
        n, h, w, c = v.nhwc()
        if g_fmt != v.fmt:
            if v.fmt == 'act':   # f32 NCHW -> act
                conv = self.empty((n, h, w, c), F16)
                if h * w == 1:
                    hip.call('sn_copy2d', g, conv, n, c, c, c, 1, 0, hip.stream())
                else:
                    hip.call('sn_transpose_batched', g, conv, n, c, h * w, c * h * w, h * w * c, h * w, c, 1, 0, hip.stream())
            else:
                conv = self.empty(v.shape, F32)
                if h * w == 1:
                    hip.call('sn_copy2d', g, conv, n, c, c, c, 0, 1, hip.stream())
                else:
                    hip.call('sn_transpose_batched', g, conv, n, h * w, c, h * w * c, c * h * w, c, h * w, 0, 1, hip.stream())
            g = conv
        if v.grad is None:
            v.grad = g
            return
        if v.fmt == 'act':
            out = self.empty(v.grad.shape, F16)
            hip.call('sn_ew_f16', v.grad, g, None, out, n * h * w, c, c, c, c, c, 1, hip.stream())
        else:
            out = self.empty(v.grad.shape, F32)
            hip.call('sn_ew_f32', v.grad, g, out, v.grad.numel(), 1, 0.0, hip.stream())
        v.grad = out

    # ------------------------------------------------------------------------------------------
    # lowering
    # ------------------------------------------------------------------------------------------
    def _lower(self):
        from . import ops
        # consumer census (for BN+ReLU fusion and f32/act decisions)
        cons = {}
        for node in self.nodes:
            for n, i in node.inputs:
                cons.setdefault((id(n), i), []).append(node)
        heads = set((id(n), i) for n, i in self.sym._heads)
        self.consumers = cons
        self.head_keys = heads       # graph outputs: read by the caller, not by a node
        var_is_param = set(self.param_names) | set(self.aux_names)
        # static format analysis: 'act' (fp16 channels-last) vs 'f32' (reference order)
        binary = ('_plus', '_minus', '_mul', 'elemwise_add')
        fmt = {}
        for node in self.nodes:
            if node.op is None:
                fmt[(id(node), 0)] = 'f32'
            elif node.op == 'Pooling' and node.attrs.get('pool_type', 'max') == 'avg':
                fmt[(id(node), 0)] = 'f32'      # global average pooling writes fp32 (N,C,1,1) (ops.PoolingStep)
            elif node.op in _PRODUCES_ACT:
                fmt[(id(node), 0)] = 'act'
            elif node.op == 'Cast':
                fmt[(id(node), 0)] = fmt[(id(node.inputs[0][0]), node.inputs[0][1])]
            elif node.op in binary:
                both = all(fmt[(id(n), i)] == 'act' for n, i in node.inputs)
                fmt[(id(node), 0)] = 'act' if both and node.op in ('_plus', 'elemwise_add') else 'f32'
            else:
                for i in range(node.num_outputs):
                    fmt[(id(node), i)] = 'f32'
        for node in self.nodes:
            if node.op is None:
                if node.name in var_is_param:
                    continue
                shp = self.shapes[('var', node.name)]
                v = Val(node.name, shp, 'f32')
                self.vals[(id(node), 0)] = v
                continue
            cls = ops.REGISTRY.get(node.op)
            if cls is None:
                raise NotImplementedError('no HIP lowering for operator %s (%s)' % (node.op, node.name))
            wants_f32 = [False] * node.num_outputs
            for i in range(node.num_outputs):
                cs = cons.get((id(node), i), [])
                if (id(node), i) in heads or any(c.op in _F32_CONSUMERS for c in cs):
                    wants_f32[i] = True
                for c in cs:   # element-wise ops follow their other operand; rois/trans slots are f32
                    if c.op in binary and fmt[(id(c), 0)] == 'f32':
                        wants_f32[i] = True
                    if c.op in ('DeformablePSROIPooling', 'pick', 'MaskRcnnTarget'):
                        slots = c.extra.get('slots') or []
                        for s, (n2, i2) in zip(slots, c.inputs):
                            if n2 is node and i2 == i and s in ('rois', 'trans', 'index', 'mask_polys', 'mask_ids'):
                                wants_f32[i] = True
            step = cls(self, node, wants_f32)
            self.steps.append(step)
        # needs_grad propagation is done by the steps at construction (they see their inputs)

    def register_param(self, name, kind='vec', fc_in=None, need_wT=False):
        shp = self.shapes[('var', name)]
        p = self.params.get(name)
        if p is None:
            p = Param(name, shp)
            p.step_index = len(self.steps)      # the step being constructed gets this index
            self.params[name] = p
        p.kind = kind
        p.fc_in = fc_in
        p.need_wT = p.need_wT or need_wT
        if kind == 'conv':
            o, i, kh, kw = shp
            p.int_shape = (o, kh * kw, i)
        elif kind == 'deconv':
            ci, co, kh, kw = shp
            p.int_shape = (kh * kw * co, 1, ci)
        elif kind == 'stem':
            kh, kw, kwp = fc_in
            p.int_shape = (shp[0], kh, kwp * 4)
        elif kind == 'fc':
            o = shp[0]
            if fc_in is not None:
                c, h, w = fc_in
                p.int_shape = (o, h * w, c)
            else:
                p.int_shape = (o, 1, int(np.prod(shp[1:])))
        else:
            p.int_shape = tuple(shp)
        p.numel = int(np.prod(p.int_shape))      # the packed stem layout is larger than the reference tensor
        p.trainable = self.for_training and not any(f == name for f in self.fixed)
        return p

    def register_aux(self, name):
        if name not in self.aux:
            shp = tuple(int(x) for x in self.shapes[('var', name)])
            t = self._share.aux.get(name) if self._share is not None else None
            if t is not None and tuple(t.shape) == shp:
                self.aux[name] = t
                self.shared_names.add(name)
            else:
                self.aux[name] = self.zeros(shp, F32)
        return self.aux[name]

    def _alloc_params(self):
        ps = list(self.params.values())
        # group so that every (lr_mult, wd_mult) class is one contiguous range -> one SGD launch each
        for p in ps:
            node = self._var_node(p.name)
            p.lr_mult = float(node.extra.get('lr_mult', 1.0)) if node is not None else 1.0
            wd_default = 1.0 if (p.name.endswith('_weight') or p.name.endswith('_gamma')) else 0.0  # mx optimizer rule
            p.wd_mult = float(node.extra.get('wd_mult', wd_default)) if node is not None else wd_default
        # fp16-region weights first (one contiguous range = the fp16 bucket of the gradient all-reduce), then by
        # (lr_mult, wd_mult) class so that every class is at most two contiguous ranges
        for p in ps:
            p.phase = 1 if (self.split_k and p.step_index < self.split_k) else 0
        train = sorted([p for p in ps if p.trainable], key=lambda q: (q.phase, not q.half_region, q.lr_mult, q.wd_mult))
        self.half_elems = sum(_pad8(p.numel) for p in train if p.half_region) if not self.split_k else 0
        # all-reduce ranges in arena order: (phase, is_half, begin, end)
        self.ar_ranges = []
        o = 0
        for p in train:
            n8 = _pad8(p.numel)
            key = (p.phase, p.half_region)
            if self.ar_ranges and tuple(self.ar_ranges[-1][:2]) == key:
                self.ar_ranges[-1][3] = o + n8
            else:
                self.ar_ranges.append([p.phase, p.half_region, o, o + n8])
            o += n8
        total = sum(_pad8(p.numel) for p in train)
        self.arena_master = self.zeros((max(total, 8),), F32)
        self.arena_grad = self.zeros((max(total, 8),), F32)
        self.arena_mom = self.zeros((max(total, 8),), F32)
        self.arena_w16 = self.zeros((max(total, 8),), F16)
        off = 0
        self.groups = []
        for p in train:
            n = p.numel
            p.offset = off
            p.master = self.arena_master[off:off + n].view(p.int_shape)
            p.grad = self.arena_grad[off:off + n].view(p.int_shape)
            p.mom = self.arena_mom[off:off + n].view(p.int_shape)
            p.w16 = self.arena_w16[off:off + n].view(p.int_shape)
            key = (p.lr_mult, p.wd_mult, (p.phase, p.half_region))
            if self.groups and self.groups[-1][0] == key:
                self.groups[-1][2] = off + _pad8(n)
            else:
                self.groups.append([key, off, off + _pad8(n)])
            off += _pad8(n)
        for p in ps:
            q = self._share.params.get(p.name) if self._share is not None else None
            if q is not None and not (q.kind == p.kind and tuple(q.int_shape) == tuple(p.int_shape) and q.master is not None
                                      and not q.trainable and not p.trainable):
                q = None
            if q is not None:
                p.master, p.w16 = q.master, q.w16
                self.shared_names.add(p.name)
            elif not p.trainable:
                p.master = self.zeros(p.int_shape, F32)
                p.w16 = self.zeros(p.int_shape, F16)
            if p.need_wT and p.kind in ('conv', 'fc', 'deconv'):
                o, t, i = p.int_shape
                if q is not None and q.wT16 is not None and tuple(q.wT16.shape) == (i, t, _pad8(o)):
                    p.wT16 = q.wT16
                else:
                    p.wT16 = self.zeros((i, t, _pad8(o)), F16)
        self.n_trainable = total

    def _choose_split(self):
        """Backward in two segments so that the gradient all-reduce of the first (heads, RPN, the late trunk) overlaps the
        second (the early trunk): the boundary is the forward step index k where the steps k.. hold about half of the
        GEMM work of the backward pass (for R101 that is the stage-3 / stage-4 boundary: 2/3 of the parameter bytes are
        complete with more than half of the backward still to run).  0 = no split."""
        work = []
        for st in self.steps:
            w = getattr(st, 'w', None)
            f = 0.0
            if w is not None and getattr(w, 'trainable', False):
                out = getattr(st, 'y', None)
                if out is not None:
                    n, h, wd, c = out.nhwc()
                    f = float(n * h * wd) * float(w.numel)
            work.append(f)
        total = sum(work)
        if total <= 0:
            return 0
        acc = 0.0
        for k in range(len(work) - 1, -1, -1):
            acc += work[k]
            if acc >= 0.5 * total:
                return k if 0 < k < len(work) - 1 else 0
        return 0

    def _mark_half_region(self):
        """Which weights the reference holds in fp16: those of the Convolution / DeformableConvolution / FullyConnected
        operators between its Cast(float16) and Cast(float32) nodes (resnet_mx_101_e2e.py:405-406, :250-252).  MXNet's
        multi-precision SGD keeps fp32 masters of them but their GRADIENTS -- what kvstore exchanges -- are fp16; the
        data-parallel all-reduce transports exactly these in fp16 (43.3 M fp16 + 30.2 M fp32 for R101)."""
        half = {}
        for node in self.nodes:
            if node.op is None:
                half[(id(node), 0)] = False
                continue
            if node.op == 'Cast':
                dt = node.attrs.get('dtype')
                try:
                    h = np.dtype(dt) == np.float16
                except TypeError:
                    h = 'float16' in str(dt)
            else:
                h = False
                for n, i in node.inputs:
                    if n.op is not None or n.name not in self.params and n.name not in self.aux:
                        h = half.get((id(n), i), False)
                        break
            for i in range(node.num_outputs):
                half[(id(node), i)] = h
            if h and node.op in ('Convolution', 'DeformableConvolution', 'FullyConnected', 'Deconvolution'):
                slots = node.extra.get('slots') or []
                for sl, (n, i) in zip(slots, node.inputs):
                    if sl in ('weight', 'bias') and n.op is None and n.name in self.params:
                        self.params[n.name].half_region = True

    def _var_node(self, name):
        idx = self.__dict__.get('_var_index')
        if idx is None:                     # (a linear search per parameter was a fifth of a test-time bind: 336 x 700 nodes)
            idx = self._var_index = {}
            for n in self.nodes:
                if n.op is None:
                    idx.setdefault(n.name, n)
        return idx.get(name)

    # ------------------------------------------------------------------------------------------
    # parameters in / out (reference layout on the host side)
    # ------------------------------------------------------------------------------------------
    def set_params(self, arg_params, aux_params=None, allow_missing=False):
        self.fold_store['__valid__'] = False        # the derived buffers (shared per Module) follow the masters again in refresh_compute_copies
        for name, p in self.params.items():
            if name in arg_params:
                a = arg_params[name]
                a = a.asnumpy() if hasattr(a, 'asnumpy') else np.asarray(a)
                if tuple(a.shape) != p.ref_shape:
                    raise ValueError('parameter %s: shape %s, expected %s' % (name, a.shape, p.ref_shape))
                t = torch.from_numpy(np.ascontiguousarray(p.to_internal(a))).to(self.device).view(p.int_shape)
                p.master.copy_(t)
            elif not allow_missing:
                raise KeyError('missing parameter %s' % name)
        for name, t in self.aux.items():
            if aux_params and name in aux_params:
                a = aux_params[name]
                a = a.asnumpy() if hasattr(a, 'asnumpy') else np.asarray(a)
                t.copy_(torch.from_numpy(np.asarray(a, np.float32)).to(self.device))
            elif not allow_missing:
                raise KeyError('missing auxiliary state %s' % name)
        self.refresh_compute_copies()

    def get_params(self):
        arg = {name: p.to_reference(p.master.detach().cpu().numpy()) for name, p in self.params.items()}
        aux = {name: t.detach().cpu().numpy().copy() for name, t in self.aux.items()}
        return arg, aux

    def refresh_compute_copies(self, only_trainable=False):
        """fp16 copies (and transposed dgrad copies) of the fp32 masters."""
        for p in self.params.values():
            if only_trainable and not p.trainable:
                continue
            if not (only_trainable and p.trainable):  # trainable w16 is written by the SGD kernel
                hip.call('sn_copy2d', p.master, p.w16, 1, p.numel, p.numel, p.numel, 1, 0, hip.stream())
        self._transpose_weights(only_trainable)
        self._bn_cache_valid = False
        for s in self.steps:
            s.params_changed(only_trainable)
        if not self.for_training:
            self.fold_store['__valid__'] = True

    def derived_buffer(self, key, shape, dtype):
        """Test time: a buffer DERIVED from the parameters alone (a BatchNorm's scale / shift from its moving statistics) -- one per
        Module like the parameters themselves (share_params): the executors of its batch shapes hold the same tensor."""
        if self.for_training:
            return self.empty(shape, dtype)
        t = self.fold_store.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = self.fold_store[key] = self.empty(shape, dtype)
        return t

    def adopt_derived(self):
        """A further batch shape of a test-time Module whose parameters are all shared with an executor that already derived
        everything from them (fp16 copies, BatchNorm scale / shift, BatchNorm-folded weights: Module._exe_for): nothing to compute --
        round 5 re-ran ~340 copies, 101 scale / shift kernels and 64 folds per new shape, 6 ms of host time each.  False = something
        is not there yet (the caller refreshes as before)."""
        if self.for_training or not self.fold_store.get('__valid__') or os.environ.get('SNIPER_ADOPT_DERIVED', '1') == '0':
            return False
        if not all(n in self.shared_names for n in list(self.params) + list(self.aux)):
            return False
        for s in self.steps:
            if not s.adopt_derived():
                return False
        return True

    def transpose_jobs(self, only_trainable):
        """(master, dst, O, T, I) of every transposed data-gradient copy: the parameters' own plus what the steps add."""
        jobs = []
        for p in self.params.values():
            if p.wT16 is None or (only_trainable and not p.trainable):
                continue
            o, t, i = p.int_shape
            if p.kind == 'fc':
                # an FC over a pooled (h, w, c) tensor is a 1x1 GEMM whose K index is the flat (hw, c) feature:
                # its data gradient needs W^T as [hw*C + c][O], not the convolution's [c][tap][O]
                t, i = 1, t * i
            jobs.append((p.master, p.wT16, o, t, i))
        for s in self.steps:
            jobs.extend(s.transpose_jobs(only_trainable))
        return jobs

    def _transpose_weights(self, only_trainable):
        """One sn_weight_transpose_batched launch over a device-resident descriptor table (built once per variant: the
        buffers never move, and the optimizer graph captures the launch)."""
        key = bool(only_trainable)
        tab = self._wt_tables.get(key)
        if tab is None:
            jobs = self.transpose_jobs(only_trainable)
            rec = np.zeros(len(jobs), dtype=np.dtype([('src', '<u8'), ('dst', '<u8'), ('O', '<i4'), ('T', '<i4'), ('I', '<i4'),
                                                     ('Opad', '<i4'), ('tile0', '<i4'), ('tiles_o', '<i4'), ('tiles_i', '<i4'),
                                                     ('pad', '<i4')]))
            assert rec.dtype.itemsize == 48
            tile0 = 0
            for k, (src, dst, o, t, i) in enumerate(jobs):
                opad = _pad8(o)
                to, ti = (opad + 63) // 64, (i + 63) // 64
                rec[k] = (src.data_ptr(), dst.data_ptr(), o, t, i, opad, tile0, to, ti, 0)
                tile0 += t * to * ti
            dev = torch.from_numpy(rec.view(np.uint8).copy()).to(self.device) if len(jobs) else None
            tab = self._wt_tables[key] = (dev, len(jobs), tile0, jobs)      # jobs kept: they own the tensors behind the pointers
        dev, n, tiles, _ = tab
        if n:
            hip.call('sn_weight_transpose_batched', dev, n, tiles, hip.stream())

    # ------------------------------------------------------------------------------------------
    # run
    # ------------------------------------------------------------------------------------------
    def load_inputs(self, inputs):
        """Copy the step's inputs into the bound (static) input buffers: captured kernels hold their addresses."""
        for node in self.nodes:
            if node.op is None and (id(node), 0) in self.vals:
                v = self.vals[(id(node), 0)]
                src = inputs[node.name]
                if hasattr(src, 'asnumpy') and not isinstance(getattr(src, '_data', None), torch.Tensor):
                    src = src.asnumpy()
                if isinstance(src, np.ndarray):
                    src = torch.from_numpy(np.ascontiguousarray(src, np.float32))
                    if self.device.type == 'cuda' and (self.for_training or src.numel() <= (1 << 18)):
                        # (training executors and small arrays only: a test-time Module holds one executor per batch shape, and
                        # three pinned copies of a 69 MB host image batch per shape would pin gigabytes of host memory)
                        # host inputs (the reference iterator's small arrays -- valid ranges, im_info -- or whole host batches) go
                        # through a ring of pinned buffers: a copy from PAGEABLE memory is synchronous, i.e. the host would wait
                        # for the step in flight before it could enqueue this one (13 ms per batch, profiles/r05_fit_path.txt)
                        ring = self.__dict__.setdefault('_pinned_inputs', {}).setdefault(node.name, {'k': 0, 'bufs': [], 'evs': []})
                        if len(ring['bufs']) < 3:
                            ring['bufs'].append(torch.empty(tuple(src.shape), dtype=torch.float32, pin_memory=True))
                            ring['evs'].append(None)
                        k = ring['k'] % len(ring['bufs'])
                        ring['k'] += 1
                        if tuple(ring['bufs'][k].shape) != tuple(src.shape):
                            ring['bufs'][k] = torch.empty(tuple(src.shape), dtype=torch.float32, pin_memory=True)
                        elif ring['evs'][k] is not None:
                            ring['evs'][k].synchronize()              # (three steps old: long complete)
                        ring['bufs'][k].copy_(src)
                        src = ring['bufs'][k]
                        ring['pending'] = k
                elif hasattr(src, '_data'):
                    src = src._data
                if tuple(src.shape) != v.shape:
                    raise ValueError('input %s has shape %s, bound shape %s' % (node.name, tuple(src.shape), v.shape))
                if v.t is None:
                    v.t = self.act_empty(v.shape, F32)
                n = v.t.numel()
                if (isinstance(src, torch.Tensor) and src.is_cuda and src.dtype == torch.float32 and src.is_contiguous()
                        and n % 8 == 0 and src.data_ptr() % 16 == 0 and v.t.data_ptr() % 16 == 0):
                    # one 16-byte-per-lane kernel per tensor: torch's device-to-device copy_ runs the 80 MB of a 20-chip step as
                    # ~60 blit launches at ~0.3 TB/s (0.29 ms per step in the kernel trace)
                    hip.call('sn_copy2d', src, v.t, 1, n, n, n, 1, 1, hip.stream())
                else:
                    v.t.copy_(src, non_blocking=True)
                    ring = getattr(self, '_pinned_inputs', {}).get(node.name)
                    if ring is not None and ring.pop('pending', None) is not None and src.is_pinned():
                        ev = torch.cuda.Event()
                        ev.record()
                        ring['evs'][(ring['k'] - 1) % len(ring['bufs'])] = ev

    def _forward_body(self):
        for v in self.vals.values():
            v.alt = None
            v.grad = None
            v.grad_group = None
        for s in self.steps:
            s.forward()
        self.outputs = [self.as_f32(self.vals[(id(n), i)]) for n, i in self.sym._heads]

    def forward(self, inputs, is_train=None):
        is_train = self.for_training if is_train is None else is_train
        self.is_train = is_train
        self.load_inputs(inputs)
        if self.use_infer_graphs and not is_train and self.device.type == 'cuda':
            if self._infer_graph is not None:
                self._infer_graph.replay()
                return self.outputs
            self._infer_calls += 1
            # the first call runs eagerly (lazy allocations, parameter packing) -- unless this executor adopted every derived buffer of its
            # Module (Module._exe_for: capture_first): then a new batch shape costs bind + capture, not bind + eager pass + capture
            if (self._infer_calls > 1 or getattr(self, 'capture_first', False)) and _CAPTURE_ALLOWED:
                import gc
                gc.collect()
                gc_was = gc.isenabled()
                gc.disable()             # (no collection -- no foreign destructor freeing memory -- on a capturing stream: _capture)
                try:
                    # capture_begin / capture_end on a side stream of this thread instead of the torch.cuda.graph context: that
                    # context synchronises the DEVICE (every other lane's queue), collects garbage and empties the allocator's cache
                    # on entry -- most of the 6 - 39 ms a capture cost per new batch shape (tools/cold_shape_probe.py).  The eager
                    # first call already made every lazy allocation, so nothing here needs freed memory.
                    g = torch.cuda.CUDAGraph()
                    if os.environ.get('SNIPER_INFER_LIGHT_CAPTURE', '1') == '0':      # (A/B: the torch context, as until round 5)
                        torch.cuda.synchronize()
                        with torch.cuda.graph(g, capture_error_mode='thread_local'):
                            self._forward_body()
                    else:
                        cur = torch.cuda.current_stream()
                        side = self.__dict__.setdefault('_capture_stream', None) or torch.cuda.Stream(device=self.device)
                        self._capture_stream = side
                        side.wait_stream(cur)
                        with torch.cuda.stream(side):
                            g.capture_begin(capture_error_mode='thread_local')
                            try:
                                self._forward_body()
                            finally:
                                g.capture_end()
                        cur.wait_stream(side)
                    g.replay()
                    self._infer_graph = g
                    if gc_was:
                        gc.enable()
                    settle_heap()
                    return self.outputs
                except Exception as e:   # noqa: BLE001 -- a capture failure means "keep running eagerly"
                    warnings.warn('sniper_amd: hipGraph capture of the inference forward failed (%r); running eagerly' % (e,))
                    self.use_infer_graphs = False
                    torch.cuda.synchronize()
                finally:
                    if gc_was:
                        gc.enable()
        self._forward_body()
        return self.outputs

    def zero_grad(self):
        self.arena_grad.zero_()

    def on_side(self, fn, keep=()):
        """Parameter-gradient launches of a step (weight gradients are queued by _wgrad, bias gradients run here)."""
        fn()

    def backward(self, segment=None):
        """segment None: the whole pass.  With a split (self.split_k): 'a' = the steps split_k.. (their parameter gradients
        are complete afterwards), 'b' = the rest; gradients of tensors that cross the boundary stay in Val.grad in between."""
        k = self.split_k if segment is not None else 0
        if segment in (None, 'a'):
            self.zero_grad()
            for v in self.vals.values():
                v.grad = None
                v.grad_group = None
        steps = self.steps if segment is None else (self.steps[k:] if segment == 'a' else self.steps[:k])
        for s in reversed(steps):
            s.backward()
        self.flush_wgrads()                       # the optimizer / all-reduce of this segment read the gradient arena next
        if segment in (None, 'b'):
            for v in self.vals.values():
                v.grad = None
                v.grad_group = None
        self._keepalive = []

    def _capture(self, fn, what, pool=None):
        """Capture fn() into a hipGraph; on failure fall back to eager execution for good.  pool: share the memory pool
        of an earlier capture that is always replayed right before this one (tensors live across the two)."""
        # No garbage collection while capturing: a collected object of an earlier executor (its hipGraphs and their private
        # memory pool, pinned buffers, events) frees device or host memory in its destructor, which is illegal on a capturing
        # stream and aborts the process from inside whatever Python line happened to allocate (seen in the -m gpu suite).
        import gc
        gc.collect()
        gc_was = gc.isenabled()
        gc.disable()
        try:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # thread_local: the RCCL watchdog thread of a multi-GPU job may query events while this thread captures
            kw = {'pool': pool} if pool is not None else {}
            with torch.cuda.graph(g, capture_error_mode='thread_local', **kw):
                fn()
            return g
        except Exception as e:   # noqa: BLE001 -- any capture failure means "run eagerly", never "stop training"
            warnings.warn('sniper_amd: hipGraph capture of the %s pass failed (%r); running eagerly' % (what, e))
            self.use_graphs = False
            torch.cuda.synchronize()
            return None
        finally:
            if gc_was:
                gc.enable()

    def forward_backward(self, inputs, between=None):
        """One training forward + backward pass (graph replay once captured).  With a backward split, `between()` runs
        after the first segment (forward + backward of the steps split_k..) has been enqueued: the data-parallel module
        starts the all-reduce of the finished gradients there, on its own stream, while the second segment runs."""
        self.is_train = True
        self.load_inputs(inputs)
        if not self.split_k:
            if self._graph_fb is not None:
                self._graph_fb.replay()
                return self.outputs

            def body():
                self._forward_body()
                self.backward()
            if self.use_graphs and self._eager_fb >= self.graph_warmup:
                self._graph_fb = self._capture(body, 'forward+backward')
                if self._graph_fb is not None:
                    self._graph_fb.replay()
                    return self.outputs
            body()
            self._eager_fb += 1
            return self.outputs

        def seg_a():
            self._forward_body()
            self.backward('a')

        def seg_b():
            self.backward('b')
        if self._graph_fb is None and self.use_graphs and self._eager_fb >= self.graph_warmup:
            ga = self._capture(seg_a, 'forward + first backward segment')
            gb = self._capture(seg_b, 'second backward segment', pool=ga.pool()) if ga is not None else None
            if ga is not None and gb is not None:
                self._graph_fb = (ga, gb)
        if self._graph_fb is not None:
            self._graph_fb[0].replay()
            if between is not None:
                between()
            self._graph_fb[1].replay()
            return self.outputs
        seg_a()
        if between is not None:
            between()
        seg_b()
        self._eager_fb += 1
        return self.outputs

    def _update_body(self, lr=None, wd=None, momentum=None, rescale_grad=None):
        for (lr_mult, wd_mult, _half), a, b in self.groups:
            hip.call('sn_sgd_mom_update_dev', self.arena_master[a:], self.arena_grad[a:], self.arena_mom[a:], self.arena_w16[a:],
                     b - a, self.hyper, float(lr_mult), float(wd_mult), hip.stream())
        self.refresh_compute_copies(only_trainable=True)

    def update(self, lr, wd, momentum, rescale_grad=1.0):
        """SGD with momentum on the fp32 masters (mx 'sgd', multi_precision; utils.py:26-33)."""
        # Hyper-parameters live on the device (the captured optimizer graph reads them).  They are written by fill kernels
        # with the value as a launch argument, and only when they change: a host-to-device copy from pageable memory makes
        # the host wait for everything already queued on the stream, i.e. for the whole previous step.
        vals = (float(lr), float(wd), float(momentum), float(rescale_grad))
        last = getattr(self, '_hyper_host', None)
        for i, v in enumerate(vals):
            if last is None or last[i] != v:
                hip.call('sn_ew_f32', None, None, self.hyper[i:], 1, 4, v, hip.stream())
        self._hyper_host = vals
        self.num_update += 1
        if self._graph_up is not None:
            self._graph_up.replay()
            return
        if self.use_graphs and self._eager_up >= self.graph_warmup:
            self._graph_up = self._capture(self._update_body, 'optimizer')
            if self._graph_up is not None:
                self._graph_up.replay()
                return
        self._update_body()
        self._eager_up += 1

    def grad_arena(self):
        """Flat fp32 gradient buffer (what the data-parallel all-reduce sums)."""
        return self.arena_grad
