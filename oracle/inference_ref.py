"""TEST INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg, tests): the reference's multi-scale aggregation on the host,
`Tester.aggregate` of lib/inference.py:152-201 -- per image and class the detections of every scale's every chip that pass the
scale's valid range (:170-190, the reference's own loops and numpy calls), then soft-NMS of each (image, class) problem under
`multiprocessing.Pool(32)` (:159, :192; `nms_worker.worker` -> `nms_wrapper.process` -> `soft_nms` -> `cpu_soft_nms(boxes,
sigma, Nt=0.3, threshold=0.001, method=2)`, lib/data_utils/data_workers.py:124-129, lib/nms/nms.py:15-40).
The soft-NMS itself is the reference's OWN compiled lib/nms/cpu_nms.pyx where oracle/build.py produced it (oracle/_ref/cpu_nms*.so,
kind "reference"), else the C restatement pinned against it (oracle/sniper_oracle.c, kind "port")."""
import glob
import importlib.util
import os

import numpy as np

from . import build

_REF = {}


def _ref_cpu_nms():
    if 'mod' not in _REF:
        mod = None
        hits = glob.glob(os.path.join(build.REF_OUT, 'cpu_nms*.so'))
        if hits:
            try:
                spec = importlib.util.spec_from_file_location('cpu_nms', hits[0])
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
            except Exception:      # noqa: BLE001 -- an unloadable artefact means "port", never a failed benchmark
                mod = None
        _REF['mod'] = mod
    return _REF['mod']


def kind():
    return 'reference' if _ref_cpu_nms() is not None else 'port'


def nms_problem(dets, sigma=0.55):
    """nms_worker.worker for TEST.NMS < 0 (soft-NMS, gaussian): lib/nms/nms.py:25-40."""
    dets = np.ascontiguousarray(dets, np.float32)
    if dets.shape[0] == 0:
        return dets
    mod = _ref_cpu_nms()
    if mod is not None:
        keep = mod.cpu_soft_nms(dets.copy(), np.float32(sigma), np.float32(0.3), np.float32(0.001), np.uint8(2))
        return np.asarray(keep)
    from . import capi
    return capi.soft_nms(dets, sigma=sigma, Nt=0.3, threshold=0.001, method=2)


def aggregate_problems(scale_cls_dets, valid_ranges, num_images, num_classes):
    """lib/inference.py:166-190, the reference's loops as written (np.where / np.intersect1d / np.vstack per chip)."""
    problems = []
    for i in range(num_images):
        for j in range(1, num_classes):
            agg_dets = np.empty((0, 5), dtype=np.float32)
            for all_cls_dets, valid_range in zip(scale_cls_dets, valid_ranges):
                for c in range(len(all_cls_dets[j][i])):
                    cls_dets = np.asarray(all_cls_dets[j][i][c], np.float32).reshape(-1, 5)
                    heights = cls_dets[:, 2] - cls_dets[:, 0]
                    widths = cls_dets[:, 3] - cls_dets[:, 1]
                    areas = widths * heights
                    lvalid_ids = np.where(areas > valid_range[0] * valid_range[0])[0] if valid_range[0] > 0 else np.arange(len(areas))
                    uvalid_ids = np.where(areas <= valid_range[1] * valid_range[1])[0] if valid_range[1] > 0 else np.arange(len(areas))
                    valid_ids = np.intersect1d(lvalid_ids, uvalid_ids)
                    cls_dets = cls_dets[valid_ids, :]
                    if cls_dets.shape[0] > 0:
                        agg_dets = np.vstack((agg_dets, cls_dets))
            problems.append(agg_dets)
    return problems


def _worker(args):
    return nms_problem(*args)


def aggregate(scale_cls_dets, valid_ranges, num_images, num_classes, sigma, pool):
    """-> all_boxes[class][image] like Tester.aggregate (without MAX_PER_IMAGE), NMS problems mapped over `pool`."""
    problems = aggregate_problems(scale_cls_dets, valid_ranges, num_images, num_classes)
    final = pool.map(_worker, [(p, sigma) for p in problems])
    all_boxes = [[[] for _ in range(num_images)] for _ in range(num_classes)]
    k = 0
    for i in range(num_images):
        for j in range(1, num_classes):
            all_boxes[j][i] = final[k]
            k += 1
    return all_boxes
