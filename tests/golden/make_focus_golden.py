"""Generates tests/golden/focus_mask_v1.npz from the REFERENCE's own anchor_worker (lib/data_utils/data_workers.py:165-192,
`gen_mask`, with TRAIN.AUTO_FOCUS and the thresholds of configs/faster/sniper_res101_e2e_autofocus.yml:107-119) for the
anchor cases of data_path_v1.npz.  Run in the build container (needs /root/reference):

    python tests/golden/make_focus_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from golden_util import anchor_case, golden  # noqa: E402
from oracle import ref_py  # noqa: E402
from sniper_amd import config as cfgmod  # noqa: E402

AF = dict(AUTO_FOCUS=True, AUTO_FOCUS_SMALL_THRESH=64, AUTO_FOCUS_DC_LOW=5, AUTO_FOCUS_DC_HIGH=90)


def main():
    ref = ref_py.load()
    cfg = cfgmod.res101_e2e()
    for k, v in AF.items():
        cfg.TRAIN[k] = v
    aw = ref.data_workers.anchor_worker(cfg, 512)
    n = int(golden()['anchor_count'])
    out = {'count': np.array(n), 'thresholds': np.array([AF['AUTO_FOCUS_DC_LOW'], AF['AUTO_FOCUS_SMALL_THRESH'],
                                                          AF['AUTO_FOCUS_DC_HIGH']], np.float32)}
    for k in range(n):
        args, seed, _ = anchor_case(k)
        np.random.seed(seed)
        w = aw.worker(args)
        out['mask_%02d' % k] = np.asarray(w[4], np.float32).reshape(-1)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'focus_mask_v1.npz')
    np.savez_compressed(path, **out)
    vals = np.concatenate([out['mask_%02d' % k] for k in range(n)])
    print('wrote', path, os.path.getsize(path), 'bytes;', n, 'cases; +1 / -1 / 0 cells:', int((vals == 1).sum()),
          int((vals == -1).sum()), int((vals == 0).sum()))


if __name__ == '__main__':
    main()
