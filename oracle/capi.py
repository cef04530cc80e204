"""ctypes wrappers over oracle/libsniper_oracle.so (TEST INFRASTRUCTURE ONLY)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libsniper_oracle.so")
        if not os.path.exists(path):
            from . import build

            build.build_restatement()
        _LIB = ctypes.CDLL(path)
        _LIB.orc_candidate_chips.restype = ctypes.c_int
        _LIB.orc_chips_generate.restype = ctypes.c_int
        _LIB.orc_nms_sorted_f32.restype = ctypes.c_int
        _LIB.orc_cpu_nms_f32.restype = ctypes.c_int
        _LIB.orc_soft_nms_f32.restype = ctypes.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def candidate_chips(width, height, chipsize, stride):
    n = _lib().orc_candidate_chips(int(width), int(height), int(chipsize), int(stride), None)
    out = np.empty((n, 4), np.float32)
    _lib().orc_candidate_chips(int(width), int(height), int(chipsize), int(stride), _p(out))
    return out


def shuffle_perm(n, seed):
    """Permutation produced by srand(seed) + libstdc++ std::random_shuffle over n ids."""
    perm = np.empty(n, np.int32)
    _lib().orc_shuffle_perm(int(n), ctypes.c_long(int(seed)), _p(perm))
    return perm


def chips_generate(boxes, width, height, chipsize, stride, perm=None, return_ids=False):
    boxes = np.ascontiguousarray(boxes, np.float32).reshape(-1, 4)
    n = boxes.shape[0]
    ncand = _lib().orc_candidate_chips(int(width), int(height), int(chipsize), int(stride), None)
    out = np.empty((max(ncand, 1), 4), np.float32)
    ids = np.empty(max(ncand, 1), np.int32)
    pp = None
    if perm is not None:
        perm = np.ascontiguousarray(perm, np.int32)
        assert perm.shape[0] == ncand
        pp = _p(perm)
    k = _lib().orc_chips_generate(_p(boxes), n, int(width), int(height), int(chipsize), int(stride), pp,
                                  _p(out), _p(ids), ncand)
    if return_ids:
        return out[:k].copy(), ids[:k].copy()
    return out[:k].copy()


def bbox_overlaps(boxes, query):
    boxes = np.ascontiguousarray(boxes, np.float64).reshape(-1, 4)
    query = np.ascontiguousarray(query, np.float64).reshape(-1, 4)
    out = np.empty((boxes.shape[0], query.shape[0]), np.float64)
    _lib().orc_bbox_overlaps_f64(_p(boxes), boxes.shape[0], _p(query), query.shape[0], _p(out))
    return out


def ignore_overlaps(boxes, query):
    boxes = np.ascontiguousarray(boxes, np.float64).reshape(-1, 4)
    query = np.ascontiguousarray(query, np.float64).reshape(-1, 4)
    out = np.empty((boxes.shape[0], query.shape[0]), np.float64)
    _lib().orc_ignore_overlaps_f64(_p(boxes), boxes.shape[0], _p(query), query.shape[0], _p(out))
    return out


def nms_sorted(boxes, thresh, max_keep=0):
    """Hard NMS on score-sorted (n, dim>=4) float32 boxes; returns kept row indices."""
    boxes = np.ascontiguousarray(boxes, np.float32)
    n, dim = boxes.shape
    keep = np.empty(max(n, 1), np.int32)
    k = _lib().orc_nms_sorted_f32(_p(boxes), n, dim, ctypes.c_float(thresh), int(max_keep), _p(keep))
    return keep[:k].copy()


def cpu_nms(dets, thresh, order=None):
    dets = np.ascontiguousarray(dets, np.float32).reshape(-1, 5)
    n = dets.shape[0]
    if order is None:
        order = dets[:, 4].argsort()[::-1]
    order = np.ascontiguousarray(order, np.int32)
    keep = np.empty(max(n, 1), np.int32)
    k = _lib().orc_cpu_nms_f32(_p(dets), n, _p(order), ctypes.c_float(thresh), _p(keep))
    return keep[:k].copy()


def soft_nms(boxes, sigma=0.5, Nt=0.3, threshold=0.001, method=2):
    """Returns the surviving (m,5) rows; like the reference, works on (a copy of) the input in place."""
    boxes = np.array(boxes, np.float32, copy=True).reshape(-1, 5)
    m = _lib().orc_soft_nms_f32(_p(boxes), boxes.shape[0], ctypes.c_float(sigma), ctypes.c_float(Nt),
                                ctypes.c_float(threshold), ctypes.c_uint(method))
    return boxes[:m].copy()


def cv_resize_linear_u8c3(im, fx, fy=None):
    """Scalar C restatement of cv2.resize(uint8 HxWx3, fx, fy, INTER_LINEAR) (sniper_oracle.c::orc_cv_resize_linear_u8c3)."""
    fy = fx if fy is None else fy
    im = np.ascontiguousarray(im, np.uint8)
    H, W, C = im.shape
    assert C == 3
    dh, dw = ctypes.c_int(), ctypes.c_int()
    _lib().orc_cv_dsize(H, W, ctypes.c_double(fx), ctypes.c_double(fy), ctypes.byref(dh), ctypes.byref(dw))
    out = np.empty((max(dh.value, 0), max(dw.value, 0), 3), np.uint8)
    _lib().orc_cv_resize_linear_u8c3.restype = ctypes.c_int
    rc = _lib().orc_cv_resize_linear_u8c3(_p(im), H, W, ctypes.c_double(fx), ctypes.c_double(fy), _p(out))
    if rc != 0:
        raise ValueError('cv::resize: empty dsize')
    return out
