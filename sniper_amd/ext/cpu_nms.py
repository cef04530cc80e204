"""`cpu_nms` extension module (lib/nms/cpu_nms.pyx).  lib/nms/nms.py:3-4 imports cpu_nms *and*
gpu_nms unconditionally, so both must exist for `import inference` to work.

cpu_nms (hard NMS, suppress when IoU >= thresh) runs on the bitmask kernel with the threshold moved
to the largest float32 strictly below `thresh` (IoU > t  <=>  IoU >= thresh for float32 IoUs).
cpu_soft_nms: sequential Gaussian soft-NMS; SURVEY.md section 8(f).1 marks its GPU version as the
next row -- until then it raises rather than silently computing on the host."""
import numpy as np

from . import gpu_nms as _g


def cpu_nms(dets, thresh):
    dets = np.ascontiguousarray(dets, np.float32)
    # largest float32 strictly below the (double) threshold: x > t  <=>  x >= thresh for float32 x
    t = np.float32(thresh)
    if float(t) >= float(thresh):
        t = np.nextafter(t, np.float32(-np.inf))
    return _g.gpu_nms(dets, float(t))


def cpu_soft_nms(boxes, sigma=0.5, Nt=0.3, threshold=0.001, method=2):
    raise NotImplementedError("soft-NMS on the GPU is SURVEY.md 8(f) item 1 (next row); not built yet")
