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
    if kind == 'nms_worker' and name == 'worker' and hasattr(owner, 'nms_wrapper') and \
            all(hasattr(owner.nms_wrapper, a) for a in ('thresh', 'sigma')):
        from ..inference import nms_wrapper
        w = nms_wrapper(owner.nms_wrapper.thresh, owner.nms_wrapper.sigma)
        return w.process_many
    return None


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

    def map(self, func, iterable, chunksize=None):
        batched = route(func)
        if batched is not None:
            self.routed_maps += 1
            return list(batched(list(iterable)))
        return super(Pool, self).map(func, iterable, chunksize)

    def map_async(self, func, iterable, chunksize=None, callback=None, error_callback=None):
        batched = route(func)
        if batched is not None:
            self.routed_maps += 1
            out = list(batched(list(iterable)))
            if callback is not None:
                callback(out)
            return _Done(out)
        return super(Pool, self).map_async(func, iterable, chunksize, callback, error_callback)

    def imap(self, func, iterable, chunksize=1):
        batched = route(func)
        if batched is not None:
            self.routed_maps += 1
            return iter(list(batched(list(iterable))))
        return super(Pool, self).imap(func, iterable, chunksize)


def _patch_module(module, base):
    """`from multiprocessing import Pool` inside a reference module -> this Pool (the module's global name only)."""
    import multiprocessing
    cur = getattr(module, 'Pool', None)
    real = getattr(multiprocessing, '_sniper_process_pool', multiprocessing.Pool)
    if cur is not None and (cur is real or cur is multiprocessing.Pool or getattr(cur, '__name__', '') == 'Pool') and cur is not Pool:
        if base == 'inference' and not (hasattr(module, 'Tester') or hasattr(module, 'imdb_detection_wrapper')):
            return          # some other module that happens to be called `inference`
        module.Pool = Pool


def install():
    """Scoped (idempotent): the reference modules that create GPU-work pools get this Pool as their `Pool`;
    `multiprocessing.Pool` is not touched."""
    from . import rank_slice
    for base in ('MNIteratorE2E', 'inference'):
        rank_slice.register_post_import(base, _patch_module)
    rank_slice.install_import_hook()
