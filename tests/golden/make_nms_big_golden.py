"""Generate tests/golden/nms_big_v1.npz from the REFERENCE's own lib/nms/cpu_nms.pyx (build container only): soft-NMS
problems of 4097 ... 12 000 boxes.  The reference has no size cap (cpu_nms.pyx:17-110); these pin the HIP kernel's
global-memory path (problems beyond the 4096 boxes one workgroup holds in LDS) and the oracle's C restatement.

Inputs come from tests/golden_util.py::nms_big_problem (seeded); stored per case: for every output row the index of the input
row whose coordinates it carries (soft-NMS permutes and drops rows, it never changes coordinates) and its float32 score.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_nms_big_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from golden_util import NMS_BIG_CASES, nms_big_expected, nms_big_problem  # noqa: E402
from make_nms_golden import load_ref_cpu_nms  # noqa: E402


def main():
    ref = load_ref_cpu_nms()
    out = {}
    for i, (n, method, thr, quant, seed) in enumerate(NMS_BIG_CASES):
        d = nms_big_problem(n, quant, seed)
        res = np.asarray(ref.cpu_soft_nms(d.copy(), np.float32(0.55), np.float32(0.3), np.float32(thr), np.uint8(method)),
                         np.float32).reshape(-1, 5)
        first = {}
        for k in range(n - 1, -1, -1):
            first[d[k, :4].tobytes()] = k
        idx = np.array([first[r[:4].tobytes()] for r in res], np.int32)
        out['idx_%d' % i] = idx
        out['score_%d' % i] = res[:, 4].copy()
        out['par_%d' % i] = np.array([0.55, 0.3, thr, method], np.float64)
        assert np.array_equal(nms_big_expected(out, i, d), res)
        print('case %d: n = %d method %d thr %g -> %d rows' % (i, n, method, thr, len(res)))
    out['n'] = np.array(len(NMS_BIG_CASES))
    path = os.path.join(HERE, 'nms_big_v1.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
