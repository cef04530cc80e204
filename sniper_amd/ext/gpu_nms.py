"""`gpu_nms` extension module (lib/nms/gpu_nms.pyx:16-50 over _nms, lib/nms/gpu_nms.hpp)."""
import numpy as np
import torch

from .. import hip

_ws = hip.Workspace()


def nms_sorted_device(d_boxes, thresh, max_keep=0):
    """d_boxes: (B,N,dim) float32 device tensor, rows sorted by descending score.
    -> keep (B,max_keep) int32, nkeep (B,) int32 (device)."""
    B, N, dim = d_boxes.shape
    mk = N if max_keep <= 0 else min(max_keep, N)
    keep = torch.empty((B, max(mk, 1)), dtype=torch.int32, device=d_boxes.device)
    nkeep = torch.empty((B,), dtype=torch.int32, device=d_boxes.device)
    ws = _ws.get(hip.query("sn_nms_workspace_bytes", B, N))
    hip.call("sn_nms_batch", d_boxes, None, B, N, dim, float(thresh), mk, ws, keep, nkeep, hip.stream())
    return keep, nkeep


def gpu_nms(dets, thresh, device_id=0):
    """dets (n,5) float32 [x1,y1,x2,y2,score] -> list of kept row indices (descending score)."""
    dets = np.ascontiguousarray(dets, np.float32)
    if dets.shape[0] == 0:
        return []
    order = dets[:, 4].argsort()[::-1].astype(np.int32)
    d = hip.dev(dets[order][None])
    keep, nkeep = nms_sorted_device(d, thresh)
    k = keep[0, :int(nkeep[0].item())].cpu().numpy()
    return list(order[k])


def gpu_nmsp(dets):
    return gpu_nms(dets, 0.7)
