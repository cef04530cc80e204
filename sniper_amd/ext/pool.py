"""`multiprocessing.Pool` for a process that owns a HIP context.

The reference forks worker processes for its data path and its test-time post-processing: `Pool(cfg.TRAIN.NUM_PROCESS)` for
`chip_worker.chip_extractor` / `box_assigner` (lib/iterators/MNIteratorE2E.py:34,52,62), `Pool(32)` for `nms_worker.worker`
(lib/inference.py:159) and `Pool(CONCURRENT_JOBS)` for one model process per GPU (lib/inference.py:459).  Their work items call
the `chips` / `bbox` / `cpu_nms` extension modules, which here are HIP kernels (sniper_amd/ext): a forked child inherits a copy of
the parent's HIP runtime state that it cannot use, and 64 children each opening their own context on the rank's GPU would cost
more than the kernels they launch.  The work is on the GPU either way -- a pool worker only enqueues it -- so the drop-in pool is
THREADS: same constructor, same `map` / `map_async` / `imap` / `apply_async` / `close` / `join` / `terminate` (it is
`multiprocessing.pool.ThreadPool`), every worker thread bound to the device that was current when the pool was created (a new
thread's current device is 0, which is wrong on every rank but the first).  `sniper_amd.ext.install()` puts it where
`from multiprocessing import Pool` finds it, so the reference's files need no edit and a harness needs no patch."""
import multiprocessing.pool


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


def install():
    """`from multiprocessing import Pool` -> the thread-backed pool (idempotent).  `multiprocessing.pool.Pool` itself -- the
    base class of ThreadPool -- is left alone."""
    import multiprocessing
    if getattr(multiprocessing.Pool, '__module__', '') != __name__:
        multiprocessing._sniper_process_pool = multiprocessing.Pool
        multiprocessing.Pool = Pool
