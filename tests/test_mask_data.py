"""CPU: the mask branch's polygon plumbing (sniper_amd/data/mask_utils.py) against the REFERENCE's anchor_worker output
(tests/golden/mask_polys_v1.npz, made by tests/golden/make_mask_golden.py) and, where the checkout is present, against
lib/data_utils/mask_utils.py directly."""
import os
import sys

import numpy as np
import pytest

from golden_util import anchor_case, golden
from sniper_amd.data import mask_utils
from sniper_amd.synthetic import make_roidb

HERE = os.path.dirname(os.path.abspath(__file__))


def test_encoded_polys_match_reference_anchor_worker():
    z = np.load(os.path.join(HERE, 'golden', 'mask_polys_v1.npz'))
    g = golden()
    roidb = make_roidb(6, seed=5, n_proposals=300, with_masks=True)
    plain = make_roidb(6, seed=5, n_proposals=300)
    assert all(np.array_equal(a['boxes'], b['boxes']) for a, b in zip(roidb, plain))      # polygons come from a separate stream
    n = int(z['count'])
    rows = trunc = 0
    for k in range(n):
        args, _, _ = anchor_case(k)
        i = int(g['anchor_%02d_src' % k][0])
        enc = mask_utils.encode_chip_masks(args[0], args[1], args[2], args[5], args[7], roidb[i]['gt_masks'])
        want = z['enc_%02d' % k]
        assert enc.dtype == np.float32 and np.array_equal(enc, want), k
        ids, boxes = mask_utils.kept_gt(args[0], args[1], args[2], args[5])
        gt = z['gt_%02d' % k]
        assert int((gt[:, 4] >= 0).sum()) == len(ids) and np.array_equal(gt[:len(ids), :4], boxes[ids])
        rows += len(ids)
        trunc += int((want[:len(ids), 1] < np.array([len(roidb[i]['gt_masks'][j]) for j in ids])).sum())
    assert rows > 150 and trunc > 0          # the 500-float budget truncated some objects


@pytest.mark.ref
def test_crop_and_encode_match_reference_mask_utils():
    sys.path.insert(0, '/root/reference/lib/data_utils')
    sys.dont_write_bytecode = True
    import importlib
    ref = importlib.import_module('mask_utils')
    assert ref.__file__.startswith('/root/reference')
    rs = np.random.RandomState(3)
    roidb = make_roidb(4, seed=9, with_masks=True)
    for r in roidb:
        polys = r['gt_masks']
        crop, scale = [float(rs.uniform(0, 100)), float(rs.uniform(0, 100)), 0, 0], float(rs.uniform(0.5, 3))
        a, b = ref.crop_polys(polys, crop, scale), mask_utils.crop_polys(polys, crop, scale)
        assert all(np.array_equal(x, y) and x.dtype == y.dtype for p, q in zip(a, b) for x, y in zip(p, q))
        cats = rs.randint(0, 80, len(polys))
        for mlen, mg in ((500, 100), (60, 3)):
            assert np.array_equal(ref.poly_encoder(a, cats, mlen, mg), mask_utils.poly_encoder(b, cats, mlen, mg))
