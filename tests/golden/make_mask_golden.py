"""Generate tests/golden/mask_polys_v1.npz from the REFERENCE ITSELF (build container only): the `encoded_polys` output of
the reference's anchor_worker.worker (lib/data_utils/data_workers.py:231-257 -> lib/data_utils/mask_utils.py) for every chip
of the golden anchor cases (tests/golden/data_path_v1.npz), with the synthetic COCO-style polygons of
sniper_amd.synthetic.make_roidb(with_masks=True).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_mask_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from oracle import ref_py  # noqa: E402
from golden_util import anchor_case, golden, ref_cfg  # noqa: E402
from sniper_amd.synthetic import make_roidb  # noqa: E402


def main():
    ref = ref_py.load()
    cfg = ref_cfg()
    aw = ref.data_workers.anchor_worker(cfg, 512)
    g = golden()
    roidb = make_roidb(6, seed=5, n_proposals=300, with_masks=True)
    out = {}
    n = int(g['anchor_count'])
    truncated = 0
    for k in range(n):
        args, seed, want = anchor_case(k)
        i = int(g['anchor_%02d_src' % k][0])
        np.random.seed(seed)
        w = aw.worker([a.copy() if hasattr(a, 'copy') else a for a in args] + [roidb[i]['gt_masks']])
        enc = np.asarray(w[-1].asnumpy() if hasattr(w[-1], 'asnumpy') else w[-1], np.float32)
        assert enc.shape == (100, 500)
        out['enc_%02d' % k] = enc
        out['gt_%02d' % k] = np.asarray(w[3].asnumpy() if hasattr(w[3], 'asnumpy') else w[3], np.float32)
        rows = enc[enc[:, 0] >= 0]
        truncated += int(sum(1 for r, m in zip(rows, [roidb[i]['gt_masks']] * len(rows)) if r[1] < 1))
    out['count'] = np.array(n)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'mask_polys_v1.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes;', n, 'chips; rows without any fitting segment:', truncated)


if __name__ == '__main__':
    main()
