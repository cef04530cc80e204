"""Generate tests/golden/data_path_v1.npz from the REFERENCE ITSELF (build container only).

Runs the reference's native chips/bbox modules (oracle/_ref) and its py2->py3 translated
data_workers (oracle/ref_py.py) on seeded synthetic inputs and records inputs + outputs.  The
fixtures are what the GPU box (no /root/reference there) checks the oracle and the HIP path against.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from oracle import ref_py  # noqa: E402
from sniper_amd import config as cfgmod  # noqa: E402
from sniper_amd.synthetic import make_roidb  # noqa: E402


def main():
    ref = ref_py.load()
    out = {}
    rs = np.random.RandomState(11)
    # ---- chips::cgenerate (lib/chips/cchips.cpp:54-177) ---------------------------------
    n_cases = 32
    for t in range(n_cases):
        W, H = int(rs.randint(300, 2100)), int(rs.randint(300, 1600))
        n = int(rs.randint(0, 60)) if t else 0
        side = np.exp(rs.uniform(np.log(4), np.log(400), size=n))
        x1 = rs.uniform(0, W - 2, size=n)
        y1 = rs.uniform(0, H - 2, size=n)
        b = np.stack((x1, y1, np.minimum(x1 + side, W - 2), np.minimum(y1 + side, H - 2)), 1).astype(np.float32)
        stride = int(rs.randint(56, 60))
        seed = 424200 + t
        ref_py.srand(seed)
        chips = np.array(ref.chips.generate(np.ascontiguousarray(b), W, H, 512, stride), np.float32).reshape(-1, 4)
        ncand = oracle.candidate_chips(W, H, 512, stride).shape[0]
        out['chips_%02d_boxes' % t] = b
        out['chips_%02d_meta' % t] = np.array([W, H, 512, stride, seed], np.int64)
        out['chips_%02d_perm' % t] = oracle.shuffle_perm(ncand, seed)  # libstdc++ random_shuffle order
        out['chips_%02d_out' % t] = chips
    out['chips_count'] = np.array(n_cases)
    # ---- bbox.pyx IoU / ignore-overlap (lib/bbox/bbox.pyx:17-95) ---------------------------
    for t in range(4):
        a = rs.uniform(0, 500, size=(rs.randint(1, 300), 4))
        a[:, 2:] += a[:, :2]
        q = np.round(rs.uniform(0, 500, size=(rs.randint(1, 60), 4)))
        q[:, 2:] += q[:, :2]
        out['iou_%d_a' % t], out['iou_%d_q' % t] = a, q
        out['iou_%d_iou' % t] = ref.bbox.bbox_overlaps_cython(a, q)
        out['iou_%d_ign' % t] = ref.bbox.ignore_overlaps_cython(a, q)
    out['iou_count'] = np.array(4)
    # ---- chip_extractor / box_assigner / anchor_worker (lib/data_utils/data_workers.py) ----
    cfg = cfgmod.res101_e2e()
    stride = 56
    np.random.seed(0)
    cw = ref.data_workers.chip_worker(cfg, 512)
    cw.chip_stride = stride
    cw.chip_generator = ref.chip_generator.chip_generator(chip_stride=stride, use_cpp=True)
    aw = ref.data_workers.anchor_worker(cfg, 512)
    roidb = make_roidb(6, seed=5, n_proposals=300)
    k = 0
    for i, r in enumerate(roidb):
        ref_py.srand(7000 + i)
        crops = cw.chip_extractor(copy.deepcopy(r))
        r1 = copy.deepcopy(r)
        r1['crops'] = crops
        ref_py.srand(9000 + i)
        props, negc, negp = cw.box_assigner(r1)
        out['img_%d_nchips' % i] = np.array(len(crops))
        out['img_%d_chips' % i] = np.array([c[0] for c in crops], np.float64).reshape(-1, 4)
        out['img_%d_chipmeta' % i] = np.array([[c[1], c[2], c[3], c[4]] for c in crops], np.float64).reshape(-1, 4)
        out['img_%d_negchips' % i] = np.array([c[0] for c in negc], np.float64).reshape(-1, 4)
        out['img_%d_negmeta' % i] = np.array([[c[1], c[2], c[3], c[4]] for c in negc], np.float64).reshape(-1, 4)
        for ci, p in enumerate(props):
            out['img_%d_props_%d' % (i, ci)] = np.asarray(p, np.int32)
        for ci, p in enumerate(negp):
            out['img_%d_negprops_%d' % (i, ci)] = np.asarray(p, np.int64)
        gtids = np.where(r['max_overlaps'] == 1)[0]
        for ci, crop in enumerate(crops):
            np.random.seed(100 + ci)
            w = aw.worker([[512, 512, crop[1]], crop[0].copy(), crop[1], props[ci], gtids, r['boxes'][gtids].copy(),
                           r['boxes'].copy(), r['max_classes'][gtids].reshape(-1, 1)])
            out['anchor_%02d_src' % k] = np.array([i, ci, 100 + ci], np.int64)
            out['anchor_%02d_label' % k] = np.asarray(w[0], np.float32).reshape(-1)
            out['anchor_%02d_tvals' % k] = np.asarray(w[1], np.float32)
            out['anchor_%02d_pids' % k] = np.stack([np.asarray(p).astype(np.int32) for p in w[2]])
            out['anchor_%02d_gt' % k] = np.asarray(w[3], np.float32)
            k += 1
    out['anchor_count'] = np.array(k)
    out['n_images'] = np.array(len(roidb))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data_path_v1.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes;', k, 'anchor cases')


if __name__ == '__main__':
    main()
