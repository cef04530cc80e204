"""Data parallel, one process per GPU, with the reference's GLOBAL-batch iterator: every rank assembles only ITS slice.

`main_train.py` sizes the batch as `len(context) * BATCH_IMAGES` (main_train.py:52-53) and hands `MNIteratorE2E` that global
size; MXNet's executor group then splits each batch over the GPUs of the one process.  With one process per GPU every rank runs
the same script, so every rank's iterator would crop, resize and anchor-label all W * B chips of a batch and the Module would keep
1 / W of them.  `patch_iterator_class` wraps `get_batch` of the iterator class: rank r assembles the chips
[cur_i + r * B, cur_i + (r + 1) * B) of every global batch and the cursor advances by the global batch, so the ranks' slices are
disjoint, together they are exactly the batch the unsliced iterator would have built (including the per-image chip cursor
`crop_idx`, which every rank advances for the WHOLE global batch), and `len(iter)`, the epoch length and the learning-rate schedule
stay those of the global batch.  (All ranks must hold the same chip database: `Module._sync_epoch` seeds
numpy from rank 0 and resets the iterator at every epoch.)  `install_import_hook` applies the wrapper to the reference's own class
the moment `iterators.MNIteratorE2E` is imported -- the file itself is untouched."""
import importlib.abc
import importlib.util
import os
import sys


def world_rank():
    return int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))


def enabled():
    return world_rank()[0] > 1 and os.environ.get('SNIPER_RANK_SLICE', '1') != '0'


def patch_iterator_class(cls):
    """Wrap cls.get_batch (MNIteratorE2E contract: cur_i, size, batch_size, _get_batch() -> batch).  Idempotent."""
    if getattr(cls, '_rank_slice_wrapped', False):
        return cls
    orig = cls.get_batch

    def get_batch(self):
        world, rank = world_rank()
        if not enabled() or getattr(self, 'rank_slice', True) is False:
            return orig(self)
        if self.cur_i >= self.size:
            return False
        gb = self.batch_size
        if gb % world:
            raise ValueError('global batch %d is not divisible by %d ranks' % (gb, world))
        lb, base = gb // world, self.cur_i
        self.cur_i, self.batch_size = base + rank * lb, lb
        try:
            self.batch = self._get_batch()
        finally:
            self.batch_size, self.cur_i = gb, base + gb
        # Per-image chip cursor (MNIteratorE2E.py:118-129): `_get_batch` picks chip `chip_order[crop_idx[image] % n]` for every
        # entry of the batch from the cursor values AT THE START of the batch and then advances the cursor of every entry's
        # image.  The sliced call advanced it for this rank's entries only; the entries the other ranks assembled advance it
        # here, so that every rank's cursors equal the unsliced iterator's after each global batch -- otherwise an image whose
        # chips fall into two ranks' slices would train its first chip twice and the next one never.
        crop_idx, inds = getattr(self, 'crop_idx', None), getattr(self, 'inds', None)
        if crop_idx is not None and inds is not None:
            for i in list(range(base, base + rank * lb)) + list(range(base + (rank + 1) * lb, base + gb)):
                crop_idx[inds[i]] = crop_idx[inds[i]] + 1
        self.rank_sliced = True            # Module.fit reads this: the batches are rank-local already
        return True
    sliced = get_batch

    def get_batch_on_worker_stream(self):
        """The reference's PrefetchingIter calls the iterator from a worker thread (lib/iterators/PrefetchingIter.py:53-67).  Its
        batch assembly here is uploads and kernel launches (routed pool maps, sniper_amd/ext/pool.py): on the consumer's stream every
        synchronous upload waits for the training step in flight.  A batch assembled OFF the main thread is therefore assembled on
        that thread's own HIP stream and carries `ready_event`; the consumer adopts it (iterators/PrefetchingIter.py::adopt_batch)."""
        st = _worker_stream()
        if st is None:
            return sliced(self)
        import torch
        with torch.cuda.stream(st):
            ok = sliced(self)
            batch = getattr(self, 'batch', None)
            if ok and batch is not None:
                ev = torch.cuda.Event()
                ev.record(st)
                try:
                    from ..iterators.PrefetchingIter import stamp_ready
                    stamp_ready(batch, ev)
                except AttributeError:
                    st.synchronize()
        return ok
    cls.get_batch = get_batch_on_worker_stream
    cls._rank_slice_wrapped = True
    return cls


_TLS = None


def _worker_stream():
    """This thread's own stream when it is NOT the main thread, has a GPU bound and still sits on the default stream (a thread that
    already switched streams -- sniper_amd's PrefetchingIter worker -- keeps its own).  SNIPER_PREFETCH_STREAM=0: never."""
    import threading
    if threading.current_thread() is threading.main_thread() or os.environ.get('SNIPER_PREFETCH_STREAM', '1') == '0':
        return None
    try:
        import torch
    except ImportError:            # pragma: no cover
        return None
    if not (torch.cuda.is_available() and torch.cuda.is_initialized()):
        return None
    global _TLS
    if _TLS is None:
        _TLS = threading.local()
    st = getattr(_TLS, 'stream', None)
    if st is None:
        if torch.cuda.current_stream() != torch.cuda.default_stream():
            return None
        st = _TLS.stream = torch.cuda.Stream()
    return st


def _patch_iterator_module(module, base):
    cls = getattr(module, base, None)
    if isinstance(cls, type) and hasattr(cls, 'get_batch') and hasattr(cls, '_get_batch'):
        patch_iterator_class(cls)


# module base name -> callbacks(module, base name) run right after a module of that name (any package prefix, outside sniper_amd)
# was executed: the reference's files are patched in memory, never on disk.  ext/pool.py registers `inference` and adds a second
# callback for `MNIteratorE2E` (their `Pool` name).
POST_IMPORT = {'MNIteratorE2E': [_patch_iterator_module]}


def register_post_import(base, fn):
    hooks = POST_IMPORT.setdefault(base, [])
    if fn not in hooks:
        hooks.append(fn)


class _Finder(importlib.abc.MetaPathFinder):
    """Post-import patch of the reference's modules (any package prefix: `iterators.MNIteratorE2E`, `lib.iterators...`)."""

    def __init__(self):
        self._busy = False

    def find_spec(self, name, path, target=None):
        if self._busy or name.rsplit('.', 1)[-1] not in POST_IMPORT or name.startswith('sniper_amd.'):
            return None
        self._busy = True
        try:
            spec = importlib.util.find_spec(name)
        except (ImportError, ValueError):
            spec = None
        finally:
            self._busy = False
        if spec is None or spec.loader is None:
            return None
        inner = spec.loader

        class Loader(importlib.abc.Loader):
            def create_module(self, sp):
                return inner.create_module(sp) if hasattr(inner, 'create_module') else None

            def exec_module(self, module):
                inner.exec_module(module)
                base = name.rsplit('.', 1)[-1]
                for fn in POST_IMPORT.get(base, []):
                    fn(module, base)
        spec.loader = Loader()
        return spec


def install_import_hook():
    if not any(isinstance(f, _Finder) for f in sys.meta_path):
        sys.meta_path.insert(0, _Finder())
    for name, mod in list(sys.modules.items()):            # already imported: patch in place
        base = name.rsplit('.', 1)[-1]
        if base in POST_IMPORT and not name.startswith('sniper_amd.') and mod is not None:
            for fn in POST_IMPORT[base]:
                fn(mod, base)


def is_rank_sliced(it):
    """True when `it` (or the iterator a PrefetchingIter wraps) assembles rank-local batches."""
    seen = [it] + list(getattr(it, 'iters', []) or []) + [getattr(it, 'iter', None)]
    for x in seen:
        if x is None:
            continue
        if getattr(x, 'rank_sliced', False):
            return True
        if enabled() and getattr(type(x), '_rank_slice_wrapped', False) and getattr(x, 'rank_slice', True) is not False:
            return True
    return False
