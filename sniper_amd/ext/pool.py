"""The reference's `multiprocessing.Pool`s, for a process that owns a HIP context.

The reference forks worker processes for its data path and its test-time post-processing: `Pool(cfg.TRAIN.NUM_PROCESS)` for
`chip_worker.chip_extractor` / `box_assigner` (lib/iterators/MNIteratorE2E.py:34,52,62), `Pool(32)` for `nms_worker.worker`
(lib/inference.py:159) and `Pool(CONCURRENT_JOBS)` for one model process per GPU (lib/inference.py:459).  Their work items call
the `chips` / `bbox` / `cpu_nms` extension modules, which here are HIP kernels (sniper_amd/ext): a forked child inherits a copy of
the parent's HIP runtime state that it cannot use, and 64 children each opening their own context on the rank's GPU would cost
more than the kernels they launch.

What the drop-in does with a `map` (round 4; until then every work item ran on a thread, i.e. the reference's per-image Python
loops one at a time under the interpreter lock, each with its own kernel launch and read-back):

  * `pool.map(chip_worker.chip_extractor, roidb_part)` / `pool.map(chip_worker.box_assigner, roidb_part)` -- the epoch chip
    database of MNIteratorE2E.reset (:34-66) -- are recognised by the bound method and run as ONE ragged launch over every
    (image, scale) unit of the part (sniper_amd/data/chip_worker.py: extract_batch / assign_batch, bit-equal per image to the
    reference's workers: tests/golden/data_path_v1.npz), with the reference worker's own scales / ranges / chip stride;
  * `pool.map(nms_worker.worker, problems)` -- Tester.aggregate's per-(image, class) soft-NMS (lib/inference.py:196) -- is ONE
    sn_soft_nms_batch launch (hard NMS: the bitmask kernel per problem);
  * anything else runs on threads bound to the creating thread's device (`multiprocessing.pool.ThreadPool` interface: same
    constructor, `map` / `map_async` / `imap` / `apply_async` / `close` / `join` / `terminate`).

Scope: `install()` replaces the name `Pool` INSIDE the reference modules that build those pools (`iterators.MNIteratorE2E`,
`inference`; a post-import hook, the files themselves are untouched).  `multiprocessing.Pool` itself is left alone: the reference's
dataset code (lib/dataset/imdb.py:81-118, coco.py:314) keeps its real process pools -- its work items are numpy / pycocotools and
never touch the GPU."""
import multiprocessing.pool
import os

import numpy as np


class _Done(object):
    """AsyncResult of a routed map (the batched call already ran)."""

    def __init__(self, value):
        self._value = value

    def get(self, timeout=None):
        return self._value

    def wait(self, timeout=None):
        return None

    def ready(self):
        return True

    def successful(self):
        return True


def _mirror_of(ref):
    """sniper_amd.data.chip_worker.chip_worker configured like the reference's chip_worker object `ref`
    (lib/data_utils/data_workers.py:374-392): same scales, valid ranges, chip size, negative-chip switch and THIS epoch's stride."""
    from ..data.chip_worker import chip_worker as mirror_cls
    m = getattr(ref, '_sniper_mirror', None)
    if m is None:
        m = mirror_cls.__new__(mirror_cls)
        m.perm_fn = None
        try:
            ref._sniper_mirror = m
        except AttributeError:
            pass
    m.valid_ranges, m.scales, m.chip_size = ref.valid_ranges, ref.scales, ref.chip_size
    m.use_neg_chips = ref.use_neg_chips
    m.res_based = isinstance(ref.scales[0], (list, tuple))
    m.chip_stride = ref.chip_stride
    return m


def route(func):
    """-> a function(list of work items) -> list of results that runs the whole map as batched GPU work, or None."""
    owner = getattr(func, '__self__', None)
    if owner is None or os.environ.get('SNIPER_POOL_ROUTE', '1') == '0':      # (=0: every work item on a thread -- A/B, tools/chipdb_bench.py)
        return None
    kind, name = type(owner).__name__, getattr(func, '__name__', '')
    if kind == 'chip_worker' and name in ('chip_extractor', 'box_assigner') and \
            all(hasattr(owner, a) for a in ('valid_ranges', 'scales', 'chip_size', 'use_neg_chips', 'chip_stride')):
        if type(owner).__module__.startswith('sniper_amd.'):
            return owner.extract_batch if name == 'chip_extractor' else owner.assign_batch
        mirror = _mirror_of(owner)
        return mirror.extract_batch if name == 'chip_extractor' else mirror.assign_batch
    if kind == 'anchor_worker' and name == 'worker' and os.environ.get('SNIPER_POOL_ROUTE_BATCH', '1') != '0' and \
            all(hasattr(owner, a) for a in ('scales', 'ratios', 'feat_stride', 'feat_height', 'batch_size', 'num_fg', 'pos_thresh',
                                            'neg_thresh', 'auto_focus')):
        return lambda items: _anchor_batch(owner, items, func)
    if kind == 'im_worker' and name == 'worker' and os.environ.get('SNIPER_POOL_ROUTE_BATCH', '1') != '0' and \
            getattr(owner, 'crop_size', None) and hasattr(owner, 'cfg'):
        return lambda items: _image_batch(owner, items, func)
    if kind == 'nms_worker' and name == 'worker' and hasattr(owner, 'nms_wrapper') and \
            all(hasattr(owner.nms_wrapper, a) for a in ('thresh', 'sigma')):
        from ..inference import nms_wrapper
        w = nms_wrapper(owner.nms_wrapper.thresh, owner.nms_wrapper.sigma)
        return w.process_many
    return None


# ---- per-batch workers of MNIteratorE2E._get_batch (lib/iterators/MNIteratorE2E.py:147,173) --------------------------------------
# `pool.map(anchor_worker.worker, worker_data)` and `thread_pool.map_async(im_worker.worker, ims)` run, in the reference, one chip
# per work item on the CPU (numpy IoU / labelling, cv2 decode + resize), return host arrays, and `_get_batch` stacks them into
# `mx.nd.zeros` tensors that the Module then uploads: 63 MB of pixels + 15 MB of labels from pageable memory every step.  Routed,
# the whole map is the mirror's batched GPU work (one sn_anchor_assign launch; one sn_im_prepare per chip from the device image
# cache) and the results STAY in HBM: the work items come back as handles to the batch tensors, and the shim's unplaced
# `mx.nd.zeros` (sniper_amd/mx/ndarray.py) is placed on the device by the first such write -- `_get_batch`'s own lines then move
# device memory.  What `_get_batch` reads in any other way still sees the reference's values: a handle materialises them
# (np.asarray / asnumpy: a copy to the host, and for the sparse (values, pids) pair the `np.where(bbox_weights == 1)` the reference
# worker returns).  Sub-sampling of the 256 RPN labels: the reference draws with numpy's global RNG per chip; the batched launch
# draws with the kernel's hash of (seed, chip, anchor) -- same rule, different draws; SNIPER_NUMPY_RNG=1 replays numpy's own draws
# instead (the labels before sub-sampling come to the host first: AnchorAssigner.numpy_replay_keys) and the routed batch is then
# bit-equal to the reference's serial map under np.random.seed.  SNIPER_POOL_ROUTE_BATCH=0: per item on threads.
class _NotRouted(Exception):
    """raised by a batched form that finds it cannot take this map after all (mask polygons, no GPU): the pool runs the work items
    on its threads as it would have without the routing"""


class _Dense(object):
    """dense device tensor a sparse reference value stands for: writes itself into a destination row"""
    _device_resident = True

    def __init__(self, t):
        self.t = t
        self.device = t.device

    def write_dense_into(self, dst):
        dst.copy_(self.t.reshape(dst.shape))


class _SparseVals(object):
    """`bbox_targets[pids]` of anchor_worker.worker (data_workers.py:354-357): the values at the positive anchors"""
    _device_resident = True

    def __init__(self, batch, i):
        self.batch, self.i = batch, i

    def __array__(self, dtype=None, copy=None):
        pid = self.batch.pids(self.i)
        v = self.batch.host('bbox_target')[self.i][pid[0], pid[1], pid[2]]
        return v if dtype is None else v.astype(dtype)


class _PidAxis(object):
    _device_resident = True

    def __init__(self, batch, i, axis):
        self.batch, self.i, self.axis = batch, i, axis

    def __len__(self):                      # (`if len(pids[0]) > 0`, MNIteratorE2E.py:190: an empty chip copies zeros)
        return 1

    def __array__(self, dtype=None, copy=None):
        return self.batch.pids(self.i)[self.axis]

    def dense_source(self, value):
        """`row[pids[0], pids[1], pids[2]] = value`: the sparse values -> the dense targets; the scalar 1.0 -> the dense weights"""
        key = 'bbox_target' if isinstance(value, _SparseVals) else 'bbox_weight'
        return _Dense(self.batch.out[key][self.i])


class _SparsePids(object):
    """`pids = np.where(bbox_weights == 1)` as the (3, n) array of anchor_worker.worker (data_workers.py:356,365)"""
    _device_resident = True

    def __init__(self, batch, i):
        self.batch, self.i = batch, i

    def __getitem__(self, axis):
        return _PidAxis(self.batch, self.i, axis)

    def __array__(self, dtype=None, copy=None):
        return np.stack(self.batch.pids(self.i)).astype(np.float32 if dtype is None else dtype)

    def asnumpy(self):
        return np.asarray(self)


class _AnchorBatch(object):
    def __init__(self, out):
        self.out, self._host, self._pids = out, {}, {}

    def host(self, key):
        if key not in self._host:
            self._host[key] = self.out[key].cpu().numpy()
        return self._host[key]

    def pids(self, i):
        if i not in self._pids:
            self._pids[i] = np.where(self.host('bbox_weight')[i] == 1)
        return self._pids[i]


def _anchor_batch(owner, items, func):
    import torch
    from .. import mx
    items = list(items)
    if not items or not torch.cuda.is_available() or any(len(d) > 8 for d in items):      # mask polygons: per item, on the pool's threads
        raise _NotRouted()
    aa = getattr(owner, '_sniper_anchor_mirror', None)
    if aa is None:
        from ..config import AttrDict
        from ..data.anchors import AnchorAssigner
        cfg = AttrDict(network=AttrDict(RPN_FEAT_STRIDE=int(owner.feat_stride), ANCHOR_RATIOS=list(owner.ratios),
                                        ANCHOR_SCALES=[float(x) for x in owner.scales]),
                       TRAIN=AttrDict(RPN_BATCH_SIZE=int(owner.batch_size), RPN_FG_FRACTION=float(owner.num_fg) / float(owner.batch_size),
                                      RPN_POSITIVE_OVERLAP=float(owner.pos_thresh), RPN_NEGATIVE_OVERLAP=float(owner.neg_thresh),
                                      AUTO_FOCUS=bool(owner.auto_focus), AUTO_FOCUS_DC_LOW=getattr(owner, 'auto_focus_dontcare_low', 0),
                                      AUTO_FOCUS_DC_HIGH=getattr(owner, 'auto_focus_dontcare_high', 0),
                                      AUTO_FOCUS_SMALL_THRESH=getattr(owner, 'auto_focus_small_thresh', 0)))
        aa = AnchorAssigner(cfg, int(owner.feat_height) * int(owner.feat_stride))
        assert aa.num_fg == int(owner.num_fg) and aa.A == int(owner.num_anchors)
        aa._seed = 0
        try:
            owner._sniper_anchor_mirror = aa
        except AttributeError:
            pass
    if os.environ.get('SNIPER_NUMPY_RNG', '0') == '1':
        # numpy's own draws (data_workers.py:327-338: npr.choice over the foreground, then the background candidates, chip by chip
        # from the global generator): the labels before sub-sampling come to the host, numpy_replay_keys consumes np.random exactly
        # as the reference's serial map would, and a second launch applies those draws -- under np.random.seed the routed batch is
        # then bit-equal to the unrouted reference call (tools/routed_batch_check.py).  Opt-in: the read-back makes the host wait for
        # the device once per batch, which the hashed draws of the default never do.
        packed = aa.pack_device(items)
        pre = aa.assign(keys=None, want_label_pre=True, packed=packed)
        keys = aa.numpy_replay_keys(pre['label_pre'].cpu().numpy())
        out = aa.assign(keys=keys, packed=packed)
    else:
        out = aa.assign(items, seed=aa._seed)
        aa._seed += 1
    batch = _AnchorBatch(out)
    focus = aa.focus_mask(items) if owner.auto_focus else None
    res = []
    for i in range(len(items)):
        r = [mx.nd.NDArray(out['label'][i:i + 1]), _SparseVals(batch, i), _SparsePids(batch, i), mx.nd.NDArray(out['gt_boxes'][i])]
        if focus is not None:
            r.append(mx.nd.NDArray(focus[i]))
        res.append(r)
    return res


def _image_batch(owner, items, func):
    import torch
    from .. import mx
    items = list(items)
    if not items or not torch.cuda.is_available():
        raise _NotRouted()
    iw = getattr(owner, '_sniper_im_mirror', None)
    if iw is None:
        from ..data.im_worker import im_worker as mirror_cls
        iw = mirror_cls(owner.cfg, crop_size=owner.crop_size)
        try:
            owner._sniper_im_mirror = iw
        except AttributeError:
            pass
    from ..data.im_worker import decode_ahead
    decode_ahead(iw, [d[0] for d in items])
    n, cs = len(items), int(owner.crop_size)
    ims = torch.zeros((n, 3, cs, cs), dtype=torch.float32, device=torch.device('cuda', torch.cuda.current_device()))
    for k, d in enumerate(items):
        iw.worker(d, ims[k])
    return [mx.nd.NDArray(ims[k]) for k in range(n)]


class Pool(multiprocessing.pool.ThreadPool):
    def __init__(self, processes=None, initializer=None, initargs=(), maxtasksperchild=None, context=None):
        device = None
        try:
            import torch
            if torch.cuda.is_available() and torch.cuda.is_initialized():
                device = torch.cuda.current_device()
        except ImportError:      # pragma: no cover - torch is the memory / stream layer of this package
            pass

        def init(*args):
            if device is not None:
                import torch
                torch.cuda.set_device(device)
            if initializer is not None:
                initializer(*args)
        super(Pool, self).__init__(processes, init, initargs)
        self.routed_maps = 0          # maps that ran as batched GPU work instead of per-item (tests, reports)

    def _routed(self, func, iterable):
        """-> (True, results) when the map ran as batched GPU work, (False, the work items) otherwise"""
        batched = route(func)
        if batched is None:
            return False, iterable
        items = list(iterable)
        try:
            out = list(batched(items))
        except _NotRouted:
            return False, items
        self.routed_maps += 1
        return True, out

    def map(self, func, iterable, chunksize=None):
        done, out = self._routed(func, iterable)
        if done:
            return out
        return super(Pool, self).map(func, out, chunksize)

    def map_async(self, func, iterable, chunksize=None, callback=None, error_callback=None):
        done, out = self._routed(func, iterable)
        if done:
            if callback is not None:
                callback(out)
            return _Done(out)
        return super(Pool, self).map_async(func, out, chunksize, callback, error_callback)

    def imap(self, func, iterable, chunksize=1):
        done, out = self._routed(func, iterable)
        if done:
            return iter(out)
        return super(Pool, self).imap(func, out, chunksize)


def _patch_module(module, base):
    """`from multiprocessing import Pool` inside a reference module -> this Pool (the module's global name only)."""
    import multiprocessing
    cur = getattr(module, 'Pool', None)
    real = getattr(multiprocessing, '_sniper_process_pool', multiprocessing.Pool)
    if cur is not None and (cur is real or cur is multiprocessing.Pool or getattr(cur, '__name__', '') == 'Pool') and cur is not Pool:
        if base == 'inference' and not (hasattr(module, 'Tester') or hasattr(module, 'imdb_detection_wrapper')):
            return          # some other module that happens to be called `inference`
        module.Pool = Pool


def _patch_thread_pool(module, base):
    """`from multiprocessing.pool import ThreadPool` inside lib/iterators/MNIteratorBase.py (:3,18: `self.thread_pool`, the pool
    `_get_batch` maps `im_worker.worker` over) -> this Pool: the same threads for everything it does not recognise, the routed
    batched form for the per-chip image workers."""
    cur = getattr(module, 'ThreadPool', None)
    if cur is multiprocessing.pool.ThreadPool and hasattr(module, 'MNIteratorBase'):
        module.ThreadPool = Pool


def install():
    """Scoped (idempotent): the reference modules that create GPU-work pools get this Pool as their `Pool`;
    `multiprocessing.Pool` is not touched."""
    from . import rank_slice
    for base in ('MNIteratorE2E', 'inference'):
        rank_slice.register_post_import(base, _patch_module)
    rank_slice.register_post_import('MNIteratorBase', _patch_thread_pool)
    rank_slice.install_import_hook()
