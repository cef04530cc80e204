"""Helpers to read tests/golden/data_path_v1.npz (made from the reference by make_golden.py)."""
import os

import numpy as np

from sniper_amd import config as cfgmod
from sniper_amd.synthetic import make_roidb

_G = None


def golden():
    global _G
    if _G is None:
        _G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'data_path_v1.npz'))
    return _G


def golden_images():
    """Rebuild the per-image structures (crops, props, roidb rows) recorded in the fixture."""
    g = golden()
    roidb = make_roidb(int(g['n_images']), seed=5, n_proposals=300)
    out = []
    for i, r in enumerate(roidb):
        n = int(g['img_%d_nchips' % i])
        meta = g['img_%d_chipmeta' % i]
        crops = [[g['img_%d_chips' % i][c], float(meta[c, 0]), int(meta[c, 1]), int(meta[c, 2]), int(meta[c, 3])]
                 for c in range(n)]
        props = [g['img_%d_props_%d' % (i, c)] for c in range(n)]
        nmeta = g['img_%d_negmeta' % i]
        neg = [[g['img_%d_negchips' % i][c], float(nmeta[c, 0]), int(nmeta[c, 1]), int(nmeta[c, 2]), int(nmeta[c, 3])]
               for c in range(len(nmeta))]
        negp = [g['img_%d_negprops_%d' % (i, c)] for c in range(len(nmeta))]
        out.append(dict(r=r, crops=crops, props=props, neg=neg, negprops=negp))
    return out


def anchor_case(k):
    """Inputs (as anchor_worker.worker takes them) + reference outputs for golden anchor case k."""
    g = golden()
    imgs = golden_images()
    i, ci, seed = [int(v) for v in g['anchor_%02d_src' % k]]
    im = imgs[i]
    r, crop = im['r'], im['crops'][ci]
    gtids = np.where(r['max_overlaps'] == 1)[0]
    args = [[512, 512, crop[1]], crop[0].copy(), crop[1], im['props'][ci], gtids, r['boxes'][gtids].copy(),
            r['boxes'].copy(), r['max_classes'][gtids].reshape(-1, 1)]
    pids = tuple(p.astype(np.int64) for p in g['anchor_%02d_pids' % k])
    tgt = np.zeros((84, 32, 32), np.float32)
    wgt = np.zeros((84, 32, 32), np.float32)
    if len(pids[0]):
        tgt[pids] = g['anchor_%02d_tvals' % k]
        wgt[pids] = 1.0
    return args, seed, (g['anchor_%02d_label' % k], tgt, wgt, g['anchor_%02d_gt' % k])


def ref_cfg():
    return cfgmod.res101_e2e()


# ---- large soft-NMS problems (tests/golden/nms_big_v1.npz, made by tests/golden/make_nms_big_golden.py) ---------------------
# The inputs are regenerated from a seed (numpy's legacy RandomState stream is frozen); the fixture holds the reference's
# output compactly: for every surviving row the index of the input row it is a copy of + its decayed float32 score.
NMS_BIG_CASES = [   # (n, method, threshold, score quantisation or None, seed)
    (4097, 2, 0.001, None, 501),      # one box beyond the LDS capacity of a workgroup
    (6000, 2, 0.001, 40, 502),        # quantised scores: ties broken by position; exact duplicates
    (6000, 1, 0.05, None, 503),       # linear method, many removals ("overwrite with the last box")
    (12000, 2, 0.001, None, 504),
]


def nms_big_problem(n, quant, seed):
    """n detection-like boxes: overlapping clusters inside a 2000 x 1400 image, scores in (0, 1]."""
    rs = np.random.RandomState(seed)
    k = max(1, n // 40)
    centres = np.stack((rs.uniform(40, 1960, k), rs.uniform(40, 1360, k)), 1)
    which = rs.randint(0, k, n)
    c = centres[which] + rs.normal(0, 12, (n, 2))
    wh = np.exp(rs.normal(np.log(70), 0.4, (n, 2)))
    s = rs.uniform(0.002, 1.0, (n, 1))
    if quant:
        s = np.ceil(s * quant) / quant
    d = np.hstack((c - wh / 2, c + wh / 2, s)).astype(np.float32)
    if quant:
        d[1:40:3, :4] = d[0, :4]          # exact duplicates: IoU 1
    return d


def nms_big_expected(z, i, d):
    """Rebuild the reference's output rows of case i from the fixture and the regenerated input d."""
    idx, sc = z['idx_%d' % i], z['score_%d' % i]
    return np.concatenate((d[idx, :4], sc[:, None]), 1).astype(np.float32)
