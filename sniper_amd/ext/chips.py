"""`chips` extension module (lib/chips/chips.pyx:19-21 -> chips::cgenerate, cchips.cpp:54-177).

`generate(boxes, width, height, chipsize, stride)` keeps the reference signature; the shuffle of
cchips.cpp:117 (libc rand()) becomes an explicit permutation: by default drawn from numpy's global
RNG (so `np.random.seed` makes runs reproducible), or passed by the caller (`perm=`) to replay a
recorded order bit-exactly.  `generate_batch` is the form the chip workers use: one launch for all
(image, scale) units of an epoch."""
import numpy as np
import torch

from .. import hip

_ws = hip.Workspace()


def num_candidates(width, height, chipsize, stride):
    return hip.query("sn_chips_num_candidates", int(width), int(height), int(chipsize), int(stride))


def generate_batch(units, perms=None):
    """units: list of (boxes (n,4) float32, width, height, chipsize, stride).
    perms: None (identity order), or list of int32 permutations (one per unit, length = candidates).
    Returns list of (k,4) float32 arrays (selected chips, greedy order)."""
    U = len(units)
    if U == 0:
        return []
    counts = [int(np.asarray(u[0]).reshape(-1, 4).shape[0]) for u in units]
    box_off = np.zeros(U + 1, np.int32)
    box_off[1:] = np.cumsum(counts)
    meta = np.array([[u[1], u[2], u[3], u[4]] for u in units], np.int32)
    ncand = [num_candidates(*m) for m in meta]
    cand_off = np.zeros(U + 1, np.int32)
    cand_off[1:] = np.cumsum(ncand)
    total = int(box_off[-1])
    if total == 0:
        return [np.zeros((0, 4), np.float32) for _ in units]
    boxes = np.concatenate([np.ascontiguousarray(u[0], np.float32).reshape(-1, 4) for u in units], 0)
    d_boxes = hip.dev(boxes)
    d_off, d_meta, d_coff = hip.dev(box_off), hip.dev(meta), hip.dev(cand_off)
    d_perm = None
    if perms is not None:
        flat = np.concatenate([np.asarray(p, np.int32).reshape(-1) if p is not None else np.arange(n, dtype=np.int32)
                               for p, n in zip(perms, ncand)])
        assert flat.shape[0] == cand_off[-1], "permutation lengths must equal the candidate counts"
        d_perm = hip.dev(flat)
    max_boxes = max(counts)
    ws = _ws.get(hip.query("sn_chips_workspace_bytes", int(cand_off[-1]), max_boxes))
    out = torch.empty((total, 4), dtype=torch.float32, device=d_boxes.device)
    ids = torch.empty((total,), dtype=torch.int32, device=d_boxes.device)
    cnt = torch.empty((U,), dtype=torch.int32, device=d_boxes.device)
    hip.call("sn_chips_generate_batch", d_boxes, d_off, d_meta, d_perm, d_coff, U, max_boxes, ws, out, ids, cnt, hip.stream())
    out, cnt = out.cpu().numpy(), cnt.cpu().numpy()
    return [out[box_off[u]:box_off[u] + cnt[u]].copy() for u in range(U)]


def generate(boxes, width, height, chipsize, stride, perm=None):
    """Reference signature (chips.pyx:19): returns list[list[float]]."""
    boxes = np.ascontiguousarray(boxes, np.float32).reshape(-1, 4)
    if boxes.shape[0] == 0:
        return []
    if perm is None:
        perm = np.random.permutation(num_candidates(width, height, chipsize, stride)).astype(np.int32)
    return generate_batch([(boxes, width, height, chipsize, stride)], [perm])[0].tolist()
