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


def allreduce_gradients(arena, dist=None, bucket_bytes=256 << 20, half_elems=0, half_buf=None):
    """In-place SUM of the flat gradient arena over all ranks (no averaging: the reference's losses carry
    grad_scale / BATCH_IMAGES per GPU and kvstore sums).  Returns the number of collectives issued.

    arena[:half_elems] are the gradients of the weights the reference holds in fp16 (Executor._mark_half_region): MXNet's
    kvstore exchanges those in fp16, and so does this -- converted into `half_buf` (persistent, fp16, >= half_elems),
    summed, converted back -- which halves the bytes on the xGMI links for 59 % of the R101 parameters.  The rest travels
    in fp32.  SNIPER_GRAD_FP16=0 keeps everything in fp32."""
    if dist is None:
        import torch.distributed as d
        dist = d if d.is_available() and d.is_initialized() else None
    if dist is None or dist.get_world_size() == 1:
        return 0
    k = 0
    n = arena.numel()
    half_elems = int(half_elems) if os.environ.get('SNIPER_GRAD_FP16', '1') != '0' else 0
    if half_elems > 0:
        h = half_buf[:half_elems] if half_buf is not None else torch.empty(half_elems, dtype=torch.float16, device=arena.device)
        _convert(arena[:half_elems], h)
        step = max(1, bucket_bytes // 2)
        for a in range(0, half_elems, step):
            dist.all_reduce(h[a:a + step])
            k += 1
        _convert(h, arena[:half_elems])
    step = max(1, bucket_bytes // arena.element_size())
    for a in range(half_elems, n, step):
        dist.all_reduce(arena[a:a + step])
        k += 1
    return k


def allreduce_ranges(arena, ranges, dist, half_buf=None, bucket_bytes=256 << 20):
    """Sum the arena ranges [(is_half, begin, end), ...] over all ranks, each in its transport dtype (see
    allreduce_gradients).  Used by the module for the two segments of a split backward pass."""
    if dist is None or dist.get_world_size() == 1:
        return 0
    use_half = os.environ.get('SNIPER_GRAD_FP16', '1') != '0'
    k = 0
    for is_half, a, b in ranges:
        if b <= a:
            continue
        if is_half and use_half:
            h = half_buf[a:b] if half_buf is not None else torch.empty(b - a, dtype=torch.float16, device=arena.device)
            _convert(arena[a:b], h)
            step = max(1, bucket_bytes // 2)
            for o in range(0, b - a, step):
                dist.all_reduce(h[o:o + step])
                k += 1
            _convert(h, arena[a:b])
        else:
            step = max(1, bucket_bytes // arena.element_size())
            for o in range(a, b, step):
                dist.all_reduce(arena[o:min(b, o + step)])
                k += 1
    return k


def _convert(src, dst):
    """fp32 <-> fp16 copy of a flat range: sn_copy2d on the device (the C ABI), torch on the host (gloo CPU tests)."""
    if src.is_cuda:
        from . import hip
        hip.call('sn_copy2d', src, dst, 1, src.numel(), src.numel(), src.numel(), 1 if src.dtype == torch.float32 else 0,
                 1 if dst.dtype == torch.float32 else 0, hip.stream())
    else:
        dst.copy_(src)


def rank_slice(a, rank, world_size):
    """Rank r's share batch[r*B:(r+1)*B] of a global batch-major array (MNIteratorBase.n_per_gpu, :21)."""
    assert a.shape[0] % world_size == 0, 'batch %d not divisible by %d ranks' % (a.shape[0], world_size)
    n = a.shape[0] // world_size
    return a[rank * n:(rank + 1) * n]
