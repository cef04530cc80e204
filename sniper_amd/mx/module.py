"""``mx.mod.Module`` over the HIP executor (main_train.py:89-94,143-146; lib/inference.py:71-74,423-428).

Data parallelism is MI355X-native: one process per GPU.  Under ``torchrun`` (WORLD_SIZE > 1) every
rank binds its LOCAL_RANK device, consumes its slice ``batch[rank*B:(rank+1)*B]`` of the iterator's
global batch (the split MNIteratorBase.n_per_gpu implies, lib/iterators/MNIteratorBase.py:21) and the
gradient arena is summed with one RCCL all-reduce per step -- the only exchange the reference's
kvstore='device' performs.  A single process asked for several contexts is refused: that is the
reference's in-process multi-GPU model, which this engine deliberately does not have.
"""
import logging
import os
import time
from collections import namedtuple

import numpy as np
import torch

from . import ndarray as nd
from .misc import save_checkpoint

BatchEndParam = namedtuple('BatchEndParams', ['epoch', 'nbatch', 'eval_metric', 'locals'])


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist
    return None


def init_distributed():
    """Join the torchrun rendezvous (RCCL backend) if launched with WORLD_SIZE > 1."""
    import torch.distributed as dist
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    if ws > 1 and not dist.is_initialized():
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count())
        dist.init_process_group(backend=os.environ.get('SNIPER_DIST_BACKEND', 'nccl'))
    return dist if ws > 1 else None


class Module(object):
    def __init__(self, symbol, data_names=('data',), label_names=('softmax_label',), logger=logging, context=None,
                 work_load_list=None, fixed_param_names=None, state_names=None):
        self.symbol = symbol
        self.data_names = list(data_names or [])
        self.label_names = list(label_names or [])
        self.logger = logger
        ctx = context if isinstance(context, (list, tuple)) else [context or nd.gpu(0)]
        self.contexts = list(ctx)
        self.fixed_param_names = list(fixed_param_names or [])
        self.binded = self.params_initialized = self.optimizer_initialized = False
        self.exe = None
        self._arg_params = self._aux_params = None
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        self.rank = int(os.environ.get('RANK', '0'))
        # True: the iterator yields the GLOBAL batch and every rank takes its slice (reference behaviour);
        # False: the iterator is already rank-local (bench.py: each rank owns an independent chip minibatch)
        self.slice_inputs = True

    # ---- properties the reference reads
    @property
    def output_names(self):
        return self.symbol.list_outputs()

    @property
    def data_shapes(self):
        return self._data_shapes

    @property
    def label_shapes(self):
        return self._label_shapes

    # ---- bind
    def _local(self, shape):
        """per-GPU shape of a batch-major input (global batch split across ranks / contexts)"""
        n = self.world if (self.world > 1 and self.slice_inputs) else 1
        if self.world == 1 and len(self.contexts) > 1:
            raise NotImplementedError(
                'one process driving %d GPUs is the reference\'s in-process model; launch one process per GPU instead: '
                'python -m torch.distributed.run --nproc-per-node %d ... (see INTEGRATION.md)' % (len(self.contexts),
                                                                                                len(self.contexts)))
        shape = tuple(shape)
        if n > 1:
            assert shape[0] % n == 0, 'batch %d not divisible by %d ranks' % (shape[0], n)
            return (shape[0] // n,) + shape[1:]
        return shape

    def bind(self, data_shapes, label_shapes=None, for_training=True, inputs_need_grad=False, force_rebind=False,
             shared_module=None, grad_req='write'):
        from ..engine.executor import Executor
        if self.binded and not force_rebind:
            return
        if self.world > 1:
            init_distributed()
        from ..engine.executor import thaw_heap
        thaw_heap()                  # (Modules dropped since the last settle_heap: their memory goes now)
        self.for_training = for_training
        self._data_shapes = [(d[0], tuple(d[1])) for d in data_shapes]
        self._label_shapes = [(d[0], tuple(d[1])) for d in (label_shapes or [])]
        shapes = {}
        for name, shp in self._data_shapes + self._label_shapes:
            shapes[name] = self._local(shp)
        args = set(self.symbol.list_arguments())
        shapes = {k: v for k, v in shapes.items() if k in args}
        # one process per GPU: rank-local device.  (Modulo the device count only matters for the single-GPU rehearsal of
        # the multi-rank control flow, SNIPER_DIST_BACKEND=gloo; a real launch has one device per local rank.)
        dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', self.contexts[0].device_id)) % torch.cuda.device_count()
                           if self.world > 1 else self.contexts[0].device_id)
        torch.cuda.set_device(dev)
        from .. import hip
        hip.use_device(dev.index)
        self._device = dev
        self._exes = {}          # test-time batches change shape (scale, chip size): one bound executor per input shape
        self.exe = self._exe_for(shapes)
        self.binded = True

    def _exe_for(self, shapes):
        from ..engine.executor import Executor
        key = tuple(sorted((k, tuple(v)) for k, v in shapes.items()))
        ex = self._exes.get(key)
        if ex is not None and not self.for_training:
            self._exes[key] = self._exes.pop(key)          # most recently used last
        if ex is None:
            # data parallel: backward in two segments, the first segment's gradients are all-reduced under the second
            ov = os.environ.get('SNIPER_OVERLAP_ALLREDUCE', '1')
            split = self.for_training and ((self.world > 1 and ov != '0') or ov == 'force')
            pool = None
            if not self.for_training and os.environ.get('SNIPER_SHARE_ACTIVATIONS', '1') != '0':
                from ..engine.executor import ActivationPool
                pool = self.__dict__.setdefault('_act_pool', None) or ActivationPool(self._device)
                self._act_pool = pool
            share = None
            if not self.for_training and self._exes and os.environ.get('SNIPER_SHARE_PARAMS', '1') != '0':
                share = next(iter(self._exes.values()))          # any bound shape: they all hold the Module's one parameter set
            ex = Executor(self.symbol, dict(shapes), for_training=self.for_training, fixed_param_names=self.fixed_param_names,
                          device=self._device, split_backward=split, act_pool=pool, share_params=share)
            if self._arg_params is not None:
                if share is not None and ex.adopt_derived():
                    # parameters AND everything derived from them are the Module's: nothing to do -- and nothing one-off left for a first,
                    # eager forward to settle, so this shape's FIRST forward is already the capture (engine/executor.py forward)
                    ex.capture_first = os.environ.get('SNIPER_CAPTURE_FIRST', '1') != '0'
                elif share is not None and all(n in ex.shared_names for n in list(ex.params) + list(ex.aux)):
                    ex.refresh_compute_copies()                  # the values are there: only this shape's derived buffers
                else:
                    ex.set_params(self._arg_params, self._aux_params)
            self._exes[key] = ex
            if not self.for_training:
                self._evict_stale(keep=ex)
                from ..engine.executor import settle_heap
                settle_heap()
        return ex

    def _evict_stale(self, keep=None):
        """Test-time executors are kept per (bucketed) batch shape, least recently used first out.  A pass over many images
        walks its area-sorted chip shapes in the same order every time, so a small first-in-first-out cache misses on EVERY
        batch once a scale has one shape more than it holds (64 images: 9 executors rebuilt per pass, 1.9 s instead of 0.3 --
        profiles/r04_infer_long_pass.txt).  The bound is memory, not a count: shapes are dropped only while this process holds
        more than SNIPER_EXE_CACHE_FRAC (default 0.6) of the card's HBM, or beyond SNIPER_EXE_CACHE executors (default 256)."""
        cap = int(os.environ.get('SNIPER_EXE_CACHE', '256'))
        frac = float(os.environ.get('SNIPER_EXE_CACHE_FRAC', '0.6'))
        total = 0
        if self._device.type == 'cuda':
            try:                          # (the runtime's own query: torch's device table can disagree with it under a
                total = torch.cuda.mem_get_info(self._device)[1]      # CUDA_VISIBLE_DEVICES it does not parse, main_test.py)
            except Exception:             # noqa: BLE001 -- no figure: the count bound alone
                total = 0

        def over():
            if len(self._exes) > cap:
                return True
            return bool(total) and torch.cuda.memory_allocated(self._device) > frac * total
        current, dropped = getattr(self, 'exe', None), False
        rotated = 0
        while len(self._exes) > 1 and over() and rotated < len(self._exes):
            key = next(iter(self._exes))
            if self._exes[key] is current or self._exes[key] is keep:
                # never the executor in use nor the one just built (`keep`, about to become self.exe): to the back of the queue
                self._exes[key] = self._exes.pop(key)
                rotated += 1
                continue
            self._exes.pop(key)
            live = set(kv for k in self._exes for kv in k)                 # (input name, shape) pairs some bound executor still takes
            for bk in [b for b in getattr(self, '_bucket_bufs', {}) if (b[0], tuple(b[1])) not in live]:
                del self._bucket_bufs[bk]              # the dropped shape's zero-padded input buffers (69 MB each at 1408 x 2048) go with it
            dropped = True
            import gc
            gc.unfreeze()                                  # (engine/executor.py::settle_heap put the executors out of the collector's reach)
            gc.collect()                                   # steps and executor reference each other: the tensors go with the cycle
        return dropped

    # ---- parameters
    def init_params(self, initializer=None, arg_params=None, aux_params=None, allow_missing=False, force_init=False,
                    allow_extra=True):
        if self.params_initialized and not force_init:
            return
        arg = dict(arg_params or {})
        aux = dict(aux_params or {})
        rs = np.random.RandomState(0)
        for name, p in self.exe.params.items():
            if name not in arg:
                if not allow_missing and arg_params is not None and initializer is None:
                    raise RuntimeError('%s is not presented' % name)
                if name.endswith('_gamma'):
                    arg[name] = np.ones(p.ref_shape, np.float32)
                elif name.endswith('_weight') and len(p.ref_shape) > 1:
                    fan_in = float(np.prod(p.ref_shape[1:]))
                    arg[name] = (rs.standard_normal(p.ref_shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)  # MSRA
                else:
                    arg[name] = np.zeros(p.ref_shape, np.float32)
        for name, t in self.exe.aux.items():
            if name not in aux:
                aux[name] = np.ones(tuple(t.shape), np.float32) if name.endswith('_var') else np.zeros(tuple(t.shape), np.float32)
        self._arg_params = {k: (v.asnumpy() if hasattr(v, 'asnumpy') else np.asarray(v)) for k, v in arg.items()}
        self._aux_params = {k: (v.asnumpy() if hasattr(v, 'asnumpy') else np.asarray(v)) for k, v in aux.items()}
        for ex in self._exes.values():
            ex.set_params(self._arg_params, self._aux_params)
        self._broadcast_params()
        self.params_initialized = True

    def _broadcast_params(self):
        """Data parallel: every replica starts from rank 0's parameters (what kvstore.init + pull does in the reference's
        Module) -- mx.random / np.random draws of the head initialisers differ between processes otherwise."""
        d = _dist()
        if d is None or d.get_world_size() == 1:
            return
        for ex in self._exes.values():
            d.broadcast(ex.arena_master, src=0)
            for p in ex.params.values():          # frozen parameters live outside the arena (Executor._alloc_params): a randomly
                if not p.trainable:               # initialised frozen weight (synthetic runs, names missing from the pretrained
                    d.broadcast(p.master, src=0)  # file) must be the same on every replica too, as kvstore init + pull makes it
            for t in ex.aux.values():
                d.broadcast(t, src=0)
            ex.refresh_compute_copies()
        arg, aux = self.exe.get_params()
        self._arg_params, self._aux_params = arg, aux

    def set_params(self, arg_params, aux_params, allow_missing=False, force_init=True, allow_extra=True):
        self.init_params(None, arg_params, aux_params, allow_missing, force_init)

    def get_params(self):
        arg, aux = self.exe.get_params()
        return {k: nd.NDArray(v) for k, v in arg.items()}, {k: nd.NDArray(v) for k, v in aux.items()}

    def save_checkpoint(self, prefix, epoch, save_optimizer_states=False):
        if self.rank != 0:
            return
        arg, aux = self.get_params()
        save_checkpoint(prefix, epoch, self.symbol, arg, aux)
        if save_optimizer_states:
            torch.save({'momentum': self.exe.arena_mom.cpu(), 'num_update': self.exe.num_update}, '%s-%04d.states' % (prefix, epoch))

    # ---- optimizer (mx 'sgd' with multi_precision; lib/train_utils/utils.py:26-33)
    def init_optimizer(self, kvstore='local', optimizer='sgd', optimizer_params=(('learning_rate', 0.01),), force_init=False):
        if optimizer != 'sgd':
            raise NotImplementedError('optimizer %r (the reference trains with sgd)' % (optimizer,))
        op = dict(optimizer_params)
        self.opt = {'lr': float(op.get('learning_rate', 0.01)), 'wd': float(op.get('wd', 0.0)),
                    'momentum': float(op.get('momentum', 0.0)), 'rescale_grad': float(op.get('rescale_grad', 1.0)),
                    'lr_scheduler': op.get('lr_scheduler')}
        if self.opt['lr_scheduler'] is not None:
            self.opt['lr_scheduler'].base_lr = self.opt['lr']
        self.optimizer_initialized = True

    # ---- compute
    def _adopt_iterator(self, train_data):
        """An iterator whose class sniper_amd.ext.rank_slice wrapped assembles only this rank's slice of every global batch:
        its batches (and provide_data) are rank-local -- nothing to slice, nothing to divide -- but the ranks must still walk
        the SAME chip database (_sync_epoch)."""
        from ..ext import rank_slice
        if self.world > 1 and rank_slice.is_rank_sliced(train_data):
            self.slice_inputs = False
            self._global_iter = True

    def _slice(self, a):
        if self.world > 1 and self.slice_inputs:
            n = a.shape[0] // self.world
            return a[self.rank * n:(self.rank + 1) * n]
        return a

    def _feed(self, data_batch):
        if getattr(data_batch, 'ready_event', None) is not None:       # assembled on the prefetch worker's stream
            from ..iterators.PrefetchingIter import adopt_batch
            adopt_batch(data_batch)
        feed = {}
        for name, arr in zip(self.data_names, data_batch.data):
            feed[name] = self._slice(arr)
        if data_batch.label is not None:
            for name, arr in zip(self.label_names, data_batch.label):
                feed[name] = self._slice(arr)
        self._labels = [self._slice(a) for a in (data_batch.label or [])]
        return {k: v for k, v in feed.items() if k in self.exe.input_names}

    def forward(self, data_batch, is_train=None):
        feed = self._feed(data_batch)
        if not self.for_training:
            feed = {k: self._bucket(v, self.__dict__.setdefault('_bucket_bufs', {}), k) for k, v in feed.items()}
            shapes = {k: tuple(v.shape) for k, v in feed.items()}
            if any(tuple(self.exe.vals_shape(k)) != s for k, s in shapes.items()):
                self.exe = self._exe_for(shapes)                       # rebind for this batch shape (MXNet reshapes too)
        self.exe.forward(feed, is_train=self.for_training if is_train is None else is_train)

    @staticmethod
    def _bucket(a, cache=None, key=None, q=64):
        """Test-time image batches are zero padded to the batch maximum anyway (MNIteratorTestAutoFocus._get_batch);
        padding H, W further up to a multiple of 64 keeps the number of distinct bound shapes small.  im_info carries
        the true sizes, so decoding and clipping are unaffected.
        cache (this Module's, per input name): the padded buffer of a bucket shape is allocated and zeroed ONCE; a batch is copied
        into its corner and only the strips a previous, larger batch left behind are cleared -- a fresh torch.zeros per batch was
        3 ms of host time (and a fill kernel over the whole buffer) per 8-image pass.  Safe per Module: every use is enqueued on the
        Module's stream before the executor copies the buffer into its bound input."""
        t = a._data if hasattr(a, '_data') else a
        if not (isinstance(t, torch.Tensor) and t.dim() == 4):
            return a
        H, W = int(t.shape[2]), int(t.shape[3])
        Hb, Wb = -(-H // q) * q, -(-W // q) * q
        if (Hb, Wb) == (H, W):
            return a
        shape = (int(t.shape[0]), int(t.shape[1]), Hb, Wb)
        if cache is None:
            out = torch.zeros(shape, dtype=t.dtype, device=t.device)
        else:
            ent = cache.get((key, shape, t.dtype, t.device))
            if ent is None:
                ent = cache[(key, shape, t.dtype, t.device)] = [torch.zeros(shape, dtype=t.dtype, device=t.device), 0, 0]
            out, h0, w0 = ent
            if h0 > H:
                out[:, :, H:h0, :].zero_()
            if w0 > W:
                out[:, :, :H, W:w0].zero_()
            ent[1], ent[2] = H, W
        out[:, :, :H, :W] = t
        return nd.NDArray(out)

    def backward(self, out_grads=None):
        self.exe.backward()

    def forward_backward(self, data_batch):
        """The training step's compute: one hipGraph replay once the executor has captured it (two with a split
        backward: the first segment's gradient all-reduce is started between them)."""
        self._comm_event = None
        self.exe.forward_backward(self._feed(data_batch), between=self._allreduce_first_segment if self.exe.split_k else None)

    def _half_buf(self):
        exe = self.exe
        if getattr(exe, '_half_buf', None) is None:
            n = max([b for _, half, _, b in exe.ar_ranges if half] or [0])
            exe._half_buf = torch.empty(max(n, 8), dtype=torch.float16, device=exe.arena_grad.device)
        return exe._half_buf

    def _allreduce_first_segment(self):
        """Gradients of the steps behind the split (heads, RPN, late trunk) are final: sum them over the ranks on a
        communication stream while the main stream runs the second backward segment (disjoint arena ranges)."""
        from ..parallel import allreduce_ranges
        d = _dist()
        if d is None or d.get_world_size() == 1:
            return
        exe = self.exe
        if getattr(self, '_comm_stream', None) is None:
            self._comm_stream = torch.cuda.Stream(device=exe.arena_grad.device)
        probe = getattr(self, '_comm_probe', None)       # bench.py: a list -> timed events of this step's overlap (else untimed events)
        ready = torch.cuda.Event(enable_timing=probe is not None)
        ready.record(torch.cuda.current_stream())
        self._comm_stream.wait_event(ready)
        with torch.cuda.stream(self._comm_stream):
            allreduce_ranges(exe.grad_arena(), [(h, a, b) for ph, h, a, b in exe.ar_ranges if ph == 0], d, self._half_buf())
            self._comm_event = torch.cuda.Event(enable_timing=probe is not None)
            self._comm_event.record(self._comm_stream)
        if probe is not None:
            probe.append({'seg2_start': ready, 'comm_end': self._comm_event})

    def update(self):
        from ..parallel import allreduce_ranges
        d = _dist()
        if d is not None and d.get_world_size() > 1:          # sum over ranks == kvstore 'device' push/pull
            exe = self.exe
            first_done = getattr(self, '_comm_event', None) is not None
            probe = getattr(self, '_comm_probe', None)
            if probe and first_done and 'seg2_end' not in probe[-1]:
                e = torch.cuda.Event(enable_timing=True)       # the second backward segment has been enqueued: its end on the main stream
                e.record(torch.cuda.current_stream())
                probe[-1]['seg2_end'] = e
            rest = [(h, a, b) for ph, h, a, b in exe.ar_ranges if not (first_done and ph == 0)]
            allreduce_ranges(exe.grad_arena(), rest, d, self._half_buf())
            if first_done:
                torch.cuda.current_stream().wait_event(self._comm_event)
                self._comm_event = None
        sched = self.opt['lr_scheduler']
        lr = sched(self.exe.num_update + 1) if sched is not None else self.opt['lr']
        self.exe.update(lr, self.opt['wd'], self.opt['momentum'], self.opt['rescale_grad'])

    def get_outputs(self, merge_multi_context=True):
        """The executor's output tensors, NOT copies: a test-time executor replays a captured forward into the same tensors, so
        what this returned for batch i is overwritten by forward(i + 1) -- asnumpy() / copy what must outlive the next forward
        (Tester copies to pinned host memory on the same stream before it launches the next batch)."""
        outs = [nd.NDArray(t) for t in self.exe.outputs]
        if merge_multi_context:
            return outs
        return [[o] for o in outs]

    def update_metric(self, eval_metric, labels):
        eval_metric.update(labels, self.get_outputs())

    def _snapshot_for_metric(self, slot):
        """Outputs and labels of the step just enqueued -> pinned host copies (two alternating sets), without waiting: a device
        clone on the step's stream (the next replay overwrites the outputs), the device -> host copy on a side stream."""
        st = self.__dict__.setdefault('_metric_state', {'stream': torch.cuda.Stream(device=self._device), 'sets': {}})
        tensors = [('o', i, t) for i, t in enumerate(self.exe.outputs)]
        host_labels = {}
        for i, a in enumerate(self._labels):
            d = a._data if isinstance(a, nd.NDArray) else a
            if isinstance(d, torch.Tensor) and d.is_cuda:
                tensors.append(('l', i, d))
            else:
                host_labels[i] = a if isinstance(a, nd.NDArray) else nd.NDArray(np.asarray(d))
        bufs = st['sets'].setdefault(slot, {})
        main = torch.cuda.current_stream(self._device)
        staged = []
        for kind, i, t in tensors:
            key = (kind, i, tuple(t.shape), t.dtype)
            ent = bufs.get(key)
            if ent is None:
                ent = bufs[key] = (torch.empty_like(t), torch.empty(tuple(t.shape), dtype=t.dtype, pin_memory=True))
            ent[0].copy_(t)                                   # on the step's stream: ordered before the next replay
            staged.append((kind, i, ent))
        cloned = torch.cuda.Event()
        cloned.record(main)
        side = st['stream']
        side.wait_event(cloned)
        with torch.cuda.stream(side):
            for _, _, (dcopy, hcopy) in staged:
                hcopy.copy_(dcopy, non_blocking=True)
            done = torch.cuda.Event()
            done.record(side)
        return staged, host_labels, done

    def _update_metric_from(self, eval_metric, snap):
        staged, host_labels, done = snap
        done.synchronize()
        outs, labels = {}, dict(host_labels)
        for kind, i, (_, hcopy) in staged:
            a = hcopy.float().numpy() if hcopy.dtype in (torch.float16, torch.bfloat16) else hcopy.numpy()
            (outs if kind == 'o' else labels)[i] = nd.NDArray(a)
        eval_metric.update([labels[i] for i in sorted(labels)], [outs[i] for i in sorted(outs)])

    def _sync_epoch(self, train_data, epoch):
        """Data parallel with a GLOBAL-batch iterator (slice_inputs): every rank builds the epoch's chip database itself, from
        numpy's global RNG (chip stride, chip permutations, negative-chip choice, index shuffle -- MNIteratorE2E.reset).  All
        ranks must build the SAME one, or rank r's slice r belongs to a different batch and the per-step all-reduces pair
        different steps: rank 0's seed is broadcast, numpy is re-seeded, the iterator reset, and the epoch length checked."""
        d = _dist()
        if d is None or d.get_world_size() == 1 or not (self.slice_inputs or getattr(self, '_global_iter', False)):
            return False
        dev = getattr(self, '_device', None) or torch.device('cpu')
        t = torch.zeros(1, dtype=torch.int64, device=dev)
        if self.rank == 0:
            t[0] = int(np.random.randint(0, 2 ** 31 - 1))
        d.broadcast(t, src=0)
        np.random.seed((int(t.item()) + epoch) % (2 ** 31 - 1))
        train_data.reset()
        n = torch.tensor([len(train_data), -len(train_data)], dtype=torch.int64, device=dev) if hasattr(train_data, '__len__') \
            else None
        if n is not None:
            d.all_reduce(n, op=d.ReduceOp.MAX)
            if int(n[0]) != -int(n[1]):
                raise RuntimeError('data-parallel ranks built epochs of different length (%d..%d batches worth of chips): the '
                                   'iterators are not synchronised' % (-int(n[1]), int(n[0])))
        return True

    # ---- training loop (main_train.py:143-146)
    def fit(self, train_data, eval_data=None, eval_metric='acc', epoch_end_callback=None, batch_end_callback=None,
            kvstore='local', optimizer='sgd', optimizer_params=(('learning_rate', 0.01),), eval_end_callback=None,
            eval_batch_end_callback=None, initializer=None, arg_params=None, aux_params=None, allow_missing=False,
            force_rebind=False, force_init=False, begin_epoch=0, num_epoch=None, validation_metric=None, monitor=None):
        assert num_epoch is not None, 'please specify number of epochs'
        self._adopt_iterator(train_data)
        self.bind(data_shapes=train_data.provide_data, label_shapes=train_data.provide_label, for_training=True,
                  force_rebind=force_rebind)
        self.init_params(initializer=initializer, arg_params=arg_params, aux_params=aux_params, allow_missing=allow_missing,
                         force_init=force_init)
        self.init_optimizer(kvstore=kvstore, optimizer=optimizer, optimizer_params=optimizer_params)
        cbs = lambda c: c if isinstance(c, (list, tuple)) else ([c] if c is not None else [])
        for epoch in range(begin_epoch, num_epoch):
            tic = time.time()
            if hasattr(eval_metric, 'reset'):
                eval_metric.reset()
            nbatch = 0
            if self._sync_epoch(train_data, epoch):
                pass                              # the iterator was re-seeded and reset on every rank
            elif epoch > begin_epoch:
                train_data.reset()
            # The reference's metrics read every output with asnumpy() (lib/train_utils/metric.py:50-369).  Under MXNet that waits
            # for the FORWARD outputs only and the backward pass runs on beneath the host's numpy; here a step is one captured
            # graph on one stream, so reading batch k's outputs right away would wait for the whole step and leave the GPU idle
            # while the host computes metrics and enqueues batch k + 1 (measured: 27.5 ms per batch for a 22.9 ms step,
            # profiles/r05_fit_path.txt).  So: outputs + labels of batch k are snapshotted on the device, copied to pinned host
            # memory on a side stream, and the metrics of batch k are updated one iteration later -- when the copy has long
            # landed and batch k + 1 is already queued.  The running metrics a batch_end_callback sees therefore cover the
            # batches up to k - 1 (the epoch-end values cover all); SNIPER_METRIC_LAG=0: update at once, as before.
            lag = os.environ.get('SNIPER_METRIC_LAG', '1') != '0' and getattr(self, '_device', None) is not None and \
                self._device.type == 'cuda' and hasattr(eval_metric, 'update')
            held = None
            for data_batch in train_data:
                self.forward_backward(data_batch)
                self.update()
                if lag:
                    snap = self._snapshot_for_metric(nbatch & 1)
                    if held is not None:
                        self._update_metric_from(eval_metric, held)
                    held = snap
                elif hasattr(eval_metric, 'update'):
                    self.update_metric(eval_metric, self._labels)
                if batch_end_callback is not None:
                    param = BatchEndParam(epoch=epoch, nbatch=nbatch, eval_metric=eval_metric, locals=locals())     # (once per batch)
                    for cb in cbs(batch_end_callback):
                        cb(param)
                nbatch += 1
            if held is not None:
                self._update_metric_from(eval_metric, held)
                held = None
            if hasattr(eval_metric, 'get_name_value'):
                for name, val in eval_metric.get_name_value():
                    self.logger.info('Epoch[%d] Train-%s=%f', epoch, name, val)
            self.logger.info('Epoch[%d] Time cost=%.3f', epoch, time.time() - tic)
            arg, aux = self.get_params()
            if self.rank == 0:                    # replicas are identical: one writer (do_checkpoint / the symbols' checkpoint_callback)
                for cb in cbs(epoch_end_callback):
                    cb(epoch, self.symbol, arg, aux)
            d = _dist()
            if d is not None and d.get_world_size() > 1:
                d.barrier()
