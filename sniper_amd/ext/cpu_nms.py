"""`cpu_nms` extension module (lib/nms/cpu_nms.pyx).  lib/nms/nms.py:3-4 imports cpu_nms *and*
gpu_nms unconditionally, so both must exist for `import inference` to work.

cpu_nms (hard NMS, suppress when IoU >= thresh) runs on the bitmask kernel with the threshold moved
to the largest float32 strictly below `thresh` (IoU > t  <=>  IoU >= thresh for float32 IoUs).
cpu_soft_nms / soft_nms_batch: the order-dependent sequential soft-NMS reproduced row for row by soft_nms_kernel
(one workgroup per (image, class) problem; sniper_amd/csrc/nms.hip); any problem size, like the reference (up to 4096 boxes out
of LDS, beyond that the same phases on the rows in global memory)."""
import numpy as np

from . import gpu_nms as _g


def cpu_nms(dets, thresh):
    dets = np.ascontiguousarray(dets, np.float32)
    # largest float32 strictly below the (double) threshold: x > t  <=>  x >= thresh for float32 x
    t = np.float32(thresh)
    if float(t) >= float(thresh):
        t = np.nextafter(t, np.float32(-np.inf))
    return _g.gpu_nms(dets, float(t))


def _soft_ws(max_n, total_rows, device):
    """Scratch for problems beyond one workgroup's LDS (the reference has no size cap, cpu_nms.pyx:17-110); None when all fit."""
    import torch

    from .. import hip
    if max_n <= hip.query('sn_soft_nms_max_boxes'):
        return None
    return torch.empty((hip.query('sn_soft_nms_workspace_bytes', total_rows),), dtype=torch.uint8, device=device)


def soft_nms_batch(problems, sigma=0.5, Nt=0.3, threshold=0.001, method=2):
    """P independent (n_p, 5) float32 problems in one launch (the per-class loop of lib/inference.py:152-230).
    -> list of (m_p, 5) arrays: surviving boxes, reference order, decayed scores."""
    import torch

    from .. import hip
    probs = [np.ascontiguousarray(b, np.float32).reshape(-1, 5) for b in problems]
    out = [None] * len(probs)
    idx = [i for i, b in enumerate(probs) if b.shape[0] > 0]
    for i, b in enumerate(probs):
        if b.shape[0] == 0:
            out[i] = b.copy()
    if not idx:
        return out
    sizes = [probs[i].shape[0] for i in idx]
    off = np.zeros(len(idx) + 1, np.int32)
    off[1:] = np.cumsum(sizes)
    d = hip.dev(np.concatenate([probs[i] for i in idx], 0))
    d_off = hip.dev(off)
    cnt = torch.empty((len(idx),), dtype=torch.int32, device=d.device)
    hip.call('sn_soft_nms_batch', d, d_off, len(idx), int(max(sizes)), int(off[-1]), float(sigma), float(Nt), float(threshold),
             int(method), _soft_ws(int(max(sizes)), int(off[-1]), d.device), cnt, hip.stream())
    h, c = d.cpu().numpy(), cnt.cpu().numpy()
    for k, i in enumerate(idx):
        out[i] = h[off[k]:off[k] + c[k]].copy()
    return out


def soft_nms_stacked(rows, sizes, sigma=0.5, Nt=0.3, threshold=0.001, method=2):
    """soft_nms_batch for problems that already are one (total, 5) float32 array + rows per problem (empty problems allowed):
    one upload, one launch, one download; the results are views of the downloaded array."""
    import torch

    from .. import hip
    rows = np.ascontiguousarray(rows, np.float32).reshape(-1, 5)
    sizes = np.asarray(sizes, np.int64).reshape(-1)
    P = len(sizes)
    if P == 0 or rows.shape[0] == 0:
        return [np.zeros((0, 5), np.float32) for _ in range(P)]
    off = np.zeros(P + 1, np.int32)
    off[1:] = np.cumsum(sizes)
    d, d_off = hip.dev(rows), hip.dev(off)
    cnt = torch.empty((P,), dtype=torch.int32, device=d.device)
    hip.call('sn_soft_nms_batch', d, d_off, P, int(sizes.max()), int(off[-1]), float(sigma), float(Nt), float(threshold), int(method),
             _soft_ws(int(sizes.max()), int(off[-1]), d.device), cnt, hip.stream())
    h, c = d.cpu().numpy(), cnt.cpu().numpy()
    starts, ends = off[:-1].tolist(), (off[:-1] + c).tolist()
    return [h[a:b] for a, b in zip(starts, ends)]


def cpu_soft_nms(boxes, sigma=0.5, Nt=0.3, threshold=0.001, method=2):
    """Reference signature (cpu_nms.pyx:17): mutates `boxes` like the original and returns the surviving rows."""
    res = soft_nms_batch([boxes], sigma, Nt, threshold, method)[0]
    if isinstance(boxes, np.ndarray) and boxes.dtype == np.float32 and boxes.ndim == 2:
        boxes[:res.shape[0]] = res          # the surviving prefix, as the in-place original leaves it
    return res
