"""Data parallelism: one process per GPU, the only exchange is the gradient sum (the reference's kvstore='device',
main_train.py:143-144).  Backend 'nccl' is RCCL over xGMI on MI355X; 'gloo' serves the CPU tests.

The gradient arena is one flat fp32 buffer, so the exchange is a handful of large all-reduces (xGMI rings are
per-link bound: few, large messages).  `bucket_bytes` splits it only to bound RCCL's staging, not for overlap: the
arena is complete when the captured forward+backward graph ends."""
import os

import torch


def world():
    return int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))


def init(backend=None):
    """Join the torchrun rendezvous when launched with WORLD_SIZE > 1; returns torch.distributed or None."""
    import torch.distributed as dist
    ws, _, local = world()
    if ws <= 1:
        return None
    if not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend)
    return dist


def allreduce_gradients(arena, dist=None, bucket_bytes=256 << 20):
    """In-place SUM of the flat gradient arena over all ranks (no averaging: the reference's losses carry
    grad_scale / BATCH_IMAGES per GPU and kvstore sums).  Returns the number of collectives issued."""
    if dist is None:
        import torch.distributed as d
        dist = d if d.is_available() and d.is_initialized() else None
    if dist is None or dist.get_world_size() == 1:
        return 0
    n = arena.numel()
    step = max(1, bucket_bytes // arena.element_size())
    k = 0
    for a in range(0, n, step):
        dist.all_reduce(arena[a:a + step])
        k += 1
    return k


def rank_slice(a, rank, world_size):
    """Rank r's share batch[r*B:(r+1)*B] of a global batch-major array (MNIteratorBase.n_per_gpu, :21)."""
    assert a.shape[0] % world_size == 0, 'batch %d not divisible by %d ranks' % (a.shape[0], world_size)
    n = a.shape[0] // world_size
    return a[rank * n:(rank + 1) * n]
