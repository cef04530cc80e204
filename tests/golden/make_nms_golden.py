"""Generate tests/golden/nms_v1.npz from the REFERENCE's own lib/nms/cpu_nms.pyx (build container only).

oracle/build.py compiles the reference's Cython source from where it lies (cpu_soft_nms, lines 17-110, verbatim; the hard
`cpu_nms` with the four buffer-type spellings of `_CPU_NMS_PATCH` that Cython 3 / LP64 need) into oracle/_ref/.  This
script runs it on seeded detection-like inputs -- overlapping clusters, quantised scores (ties), empty and single-box
problems, all three soft-NMS methods -- and records inputs + outputs; the GPU box (no /root/reference there) checks the
oracle and the HIP kernels against them.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_nms_golden.py
"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import build  # noqa: E402


def load_ref_cpu_nms():
    path = build.build_reference_cpu_nms()
    spec = importlib.util.spec_from_file_location('cpu_nms', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def problem(rs, n, quant=None):
    """n boxes in a few overlapping clusters inside a 512 chip, scores in (0, 1]."""
    k = max(1, n // 12)
    centres = rs.uniform(40, 470, (k, 2))
    which = rs.randint(0, k, n)
    c = centres[which] + rs.normal(0, 9, (n, 2))
    wh = np.exp(rs.normal(np.log(60), 0.35, (n, 2)))
    s = rs.uniform(0.002, 1.0, (n, 1))
    if quant:
        s = np.ceil(s * quant) / quant
    d = np.hstack((c - wh / 2, c + wh / 2, s)).astype(np.float32)
    if n > 3 and quant:
        d[1, :4] = d[0, :4]          # exact duplicates: IoU 1, tie-breaking by position
    return d


def main():
    ref = load_ref_cpu_nms()
    rs = np.random.RandomState(77)
    out = {}
    cases = []
    for n in (0, 1, 2, 7, 40, 150, 400, 1000):
        for method in (1, 2, 0):
            for thr, quant in ((0.001, None), (0.05, 20)):
                cases.append((n, method, thr, quant))
    for i, (n, method, thr, quant) in enumerate(cases):
        d = problem(rs, n, quant)
        res = np.asarray(ref.cpu_soft_nms(d.copy(), 0.55, 0.3, thr, method), np.float32).reshape(-1, 5) if n else d.copy()
        out['soft_in_%d' % i] = d
        out['soft_par_%d' % i] = np.array([0.55, 0.3, thr, method], np.float64)
        out['soft_out_%d' % i] = res
    out['soft_n'] = np.array(len(cases))
    hard = []
    for n in (1, 5, 60, 300, 900):
        for thr in (0.3, 0.5):
            hard.append((n, thr))
    for i, (n, thr) in enumerate(hard):
        d = problem(rs, n, 50 if i % 2 else None)
        out['hard_in_%d' % i] = d
        out['hard_thr_%d' % i] = np.array(thr)
        out['hard_keep_%d' % i] = np.asarray(ref.cpu_nms(d.copy(), thr), np.int32)
    out['hard_n'] = np.array(len(hard))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'nms_v1.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes;', len(cases), 'soft +', len(hard), 'hard cases')


if __name__ == '__main__':
    main()
