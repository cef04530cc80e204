"""`bbox` extension module (lib/bbox/bbox.pyx:17-95) over sn_iou_f64."""
import numpy as np
import torch

from .. import hip


def _run(boxes, query_boxes, mode):
    boxes = np.ascontiguousarray(boxes, dtype=np.float64).reshape(-1, 4)
    query_boxes = np.ascontiguousarray(query_boxes, dtype=np.float64).reshape(-1, 4)
    N, K = boxes.shape[0], query_boxes.shape[0]
    if N == 0 or K == 0:
        return np.zeros((N, K), dtype=np.float64)
    b, q = hip.dev(boxes), hip.dev(query_boxes)
    out = torch.empty((N, K), dtype=torch.float64, device=b.device)
    hip.call("sn_iou_f64", b, N, q, K, out, mode, hip.stream())
    return out.cpu().numpy()


def bbox_overlaps_cython(boxes, query_boxes):
    """(N,4) f64, (K,4) f64 -> (N,K) f64 IoU with the +1 pixel convention."""
    return _run(boxes, query_boxes, 0)


def ignore_overlaps_cython(boxes, query_boxes):
    """(N,4), (K,4) -> (N,K) intersection / query-box area."""
    return _run(boxes, query_boxes, 1)
