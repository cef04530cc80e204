"""The oracle against the committed golden vectors (generated from the reference itself by
tests/golden/make_golden.py) and hand-derived known-answer cases.  CPU only; runs everywhere."""
import copy

import numpy as np

import oracle
from oracle import data_path
from golden_util import anchor_case, golden, golden_images, ref_cfg


def test_chips_golden():
    g = golden()
    n = int(g['chips_count'])
    assert n >= 16
    tot = 0
    for t in range(n):
        W, H, cs, stride, seed = [int(v) for v in g['chips_%02d_meta' % t]]
        perm = g['chips_%02d_perm' % t]
        # the libc/libstdc++ shuffle restatement must reproduce the recorded permutation
        assert np.array_equal(oracle.shuffle_perm(len(perm), seed), perm)
        got = oracle.chips_generate(g['chips_%02d_boxes' % t], W, H, cs, stride, perm)
        assert np.array_equal(got, g['chips_%02d_out' % t]), t
        tot += len(got)
    assert tot > 50
    # empty box list -> no chips (cchips.cpp:56-57)
    assert oracle.chips_generate(np.zeros((0, 4), np.float32), 800, 600, 512, 56).shape == (0, 4)


def test_candidate_enumeration_counts():
    # C = 3 + nx*ny + ny + nx with nx = ceil((W-512)/s)^+
    for W, H, s in ((1920, 1440, 56), (512, 384, 57), (700, 300, 59), (513, 513, 58)):
        nx = max(0, -(-(W - 512) // s))
        ny = max(0, -(-(H - 512) // s))
        c = oracle.candidate_chips(W, H, 512, s)
        assert c.shape[0] == 3 + nx * ny + nx + ny
    # coarsest-scale case: max side 512 -> only the three (identical) corner chips
    c = oracle.candidate_chips(512, 384, 512, 56)
    assert c.shape[0] == 3 and np.array_equal(c[0], c[2])
    # quirk: y2 of the first corner is min(chipsize, height-1), not chipsize-1
    assert oracle.candidate_chips(2000, 1500, 512, 56)[0, 3] == 512


def test_iou_golden():
    g = golden()
    for t in range(int(g['iou_count'])):
        a, q = g['iou_%d_a' % t], g['iou_%d_q' % t]
        assert np.array_equal(oracle.bbox_overlaps(a, q), g['iou_%d_iou' % t])
        assert np.array_equal(oracle.ignore_overlaps(a, q), g['iou_%d_ign' % t])


def _perm_fn(seed):
    state = {'first': True}

    def fn(n):
        p = oracle.shuffle_perm(n, seed if state['first'] else -1)
        state['first'] = False
        return p

    return fn


def test_chip_extractor_box_assigner_golden():
    cfg = ref_cfg()
    for i, im in enumerate(golden_images()):
        r = im['r']
        got = data_path.chip_extractor(r, cfg.TRAIN.SCALES, cfg.TRAIN.VALID_RANGES, 512, 56, _perm_fn(7000 + i))
        assert len(got) == len(im['crops'])
        for a, b in zip(got, im['crops']):
            assert np.array_equal(a[0], b[0]) and a[1:] == b[1:]
        r2 = copy.deepcopy(r)
        r2['crops'] = got
        p, nc, npp = data_path.box_assigner(r2, cfg.TRAIN.SCALES, cfg.TRAIN.VALID_RANGES, 512, 56, True,
                                            _perm_fn(9000 + i))
        assert all(np.array_equal(a, b) for a, b in zip(p, im['props'])) and len(p) == len(im['props'])
        assert len(nc) == len(im['neg'])
        for a, b in zip(nc, im['neg']):
            assert np.array_equal(a[0], b[0]) and a[1:] == b[1:]
        assert all(np.array_equal(a, b) for a, b in zip(npp, im['negprops']))


def test_anchor_target_golden():
    cfg = ref_cfg()
    at = data_path.AnchorTarget(512, 16, cfg.network.ANCHOR_RATIOS, cfg.network.ANCHOR_SCALES)
    assert at.A == 21 and at.anchors.shape == (21504, 4)
    g = golden()
    nfg = 0
    for k in range(int(g['anchor_count'])):
        args, seed, want = anchor_case(k)
        np.random.seed(seed)
        got = at(*args)
        for a, b in zip(got, want):
            assert np.array_equal(a, b), k
        # invariants of data_workers.py:327-338: <=128 fg, <=256 sampled
        assert (got[0] == 1).sum() <= 128 and (got[0] >= 0).sum() <= 256
        nfg += int((got[0] == 1).sum())
    assert nfg > 0


def _np_nms(dets, thresh):
    """The reference's numpy nms() (lib/nms/nms.py:90-127) -- restated here only as a cross-check of
    the C oracle on sorted input."""
    x1, y1, x2, y2 = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3]
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = np.arange(len(dets))
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(i)
        xx1 = np.maximum(x1[i], x1[order[1:]])
        yy1 = np.maximum(y1[i], y1[order[1:]])
        xx2 = np.minimum(x2[i], x2[order[1:]])
        yy2 = np.minimum(y2[i], y2[order[1:]])
        w = np.maximum(np.float32(0.0), xx2 - xx1 + 1)
        h = np.maximum(np.float32(0.0), yy2 - yy1 + 1)
        inter = w * h
        ovr = inter / (areas[i] + areas[order[1:]] - inter)
        order = order[np.where(ovr <= thresh)[0] + 1]
    return np.array(keep)


def _rand_dets(rs, n, span=600):
    c = rs.uniform(0, span, size=(n, 2))
    wh = np.exp(rs.uniform(np.log(8), np.log(300), size=(n, 2)))
    d = np.concatenate((c - wh / 2, c + wh / 2, rs.uniform(0, 1, size=(n, 1))), 1).astype(np.float32)
    return d[np.argsort(-d[:, 4], kind='stable')]


def test_nms_sorted_equals_numpy_nms():
    rs = np.random.RandomState(4)
    for n in (1, 2, 63, 64, 65, 500, 2000):
        d = _rand_dets(rs, n)
        for th in (0.3, 0.5, 0.7):
            assert np.array_equal(oracle.nms_sorted(d, th), _np_nms(d, np.float32(th)))
    assert oracle.nms_sorted(np.zeros((0, 5), np.float32), 0.5).size == 0
    # max_keep truncates the survivor list (rpn_post_nms_top_n)
    d = _rand_dets(rs, 1000)
    full = oracle.nms_sorted(d, 0.7)
    assert np.array_equal(oracle.nms_sorted(d, 0.7, 50), full[:50])
    # idempotence: NMS of the survivors keeps everything
    assert len(oracle.nms_sorted(d[full], 0.7)) == len(full)


def test_nms_tie_rule_known_answers():
    # two boxes with IoU exactly 1/3: bitmask/numpy NMS suppresses only if IoU > thresh,
    # cpu_nms (cpu_nms.pyx:160) suppresses if ovr >= thresh
    a = [0, 0, 9, 9, 0.9]      # area 100
    b = [5, 0, 14, 9, 0.8]     # inter 50 -> iou 50/150
    d = np.array([a, b], np.float32)
    th = np.float32(50.0) / np.float32(150.0)
    assert list(oracle.nms_sorted(d, float(th))) == [0, 1]
    assert list(oracle.cpu_nms(d, float(th))) == [0]
    assert list(oracle.nms_sorted(d, 0.3)) == [0]


def test_soft_nms_known_answers():
    # single box: untouched.  two identical boxes (ov == 1): second decays by exp(-1/sigma)
    one = np.array([[0, 0, 9, 9, 0.5]], np.float32)
    assert np.array_equal(oracle.soft_nms(one, 0.5), one)
    two = np.array([[0, 0, 9, 9, 0.5], [0, 0, 9, 9, 0.9]], np.float32)
    out = oracle.soft_nms(two, sigma=0.5)
    assert out.shape == (2, 5) and out[0, 4] == np.float32(0.9)
    assert out[1, 4] == np.float32(np.float32(np.exp(-1.0 / 0.5)) * np.float32(0.5))
    # decay below threshold removes the box (swap-with-last)
    three = np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 9, 0.005], [100, 100, 120, 120, 0.3]], np.float32)
    out = oracle.soft_nms(three, sigma=0.5, threshold=0.001)
    assert out.shape[0] == 2 and set(np.round(out[:, 4], 3)) == {np.float32(0.9), np.float32(0.3)}
    # disjoint boxes: pure selection sort by score
    rs = np.random.RandomState(0)
    n = 20
    d = np.zeros((n, 5), np.float32)
    d[:, 0] = np.arange(n) * 50
    d[:, 2] = d[:, 0] + 10
    d[:, 3] = 10
    d[:, 4] = rs.uniform(0.1, 1, n)
    out = oracle.soft_nms(d, 0.55)
    assert np.array_equal(out[:, 4], np.sort(d[:, 4])[::-1])


def _nms_golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'nms_v1.npz'))


def test_soft_and_hard_nms_reference_golden():
    """tests/golden/nms_v1.npz holds inputs and outputs of the REFERENCE's compiled lib/nms/cpu_nms.pyx
    (tests/golden/make_nms_golden.py): 48 soft-NMS problems (3 methods, ties, duplicates, removals, n = 0..1000) and 10
    hard cpu_nms problems.  Rows, order and float32 scores must be identical."""
    z = _nms_golden()
    for i in range(int(z['soft_n'])):
        d = z['soft_in_%d' % i]
        sigma, Nt, thr, method = z['soft_par_%d' % i]
        want = z['soft_out_%d' % i]
        got = oracle.soft_nms(d.copy(), sigma, Nt, thr, int(method)) if d.shape[0] else d
        assert got.shape == want.shape and np.array_equal(got, want), i
    for i in range(int(z['hard_n'])):
        assert np.array_equal(oracle.cpu_nms(z['hard_in_%d' % i], float(z['hard_thr_%d' % i])), z['hard_keep_%d' % i]), i


def test_soft_nms_large_problems_reference_golden():
    """tests/golden/nms_big_v1.npz: the reference's compiled cpu_soft_nms on 4097 ... 12 000 boxes (it has no size cap,
    cpu_nms.pyx:17-110).  Rows, order and float32 scores of the C restatement must be identical."""
    import os
    from golden_util import NMS_BIG_CASES, nms_big_expected, nms_big_problem
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'nms_big_v1.npz'))
    assert int(z['n']) == len(NMS_BIG_CASES)
    for i, (n, method, thr, quant, seed) in enumerate(NMS_BIG_CASES):
        d = nms_big_problem(n, quant, seed)
        sigma, Nt, thr_, method_ = z['par_%d' % i]
        want = nms_big_expected(z, i, d)
        got = oracle.soft_nms(d.copy(), sigma, Nt, thr_, int(method_))
        assert got.shape == want.shape and np.array_equal(got, want), i
        assert 0 < len(want) < n          # removals happened: the "overwrite with the last box" path is exercised


def test_focus_mask_golden():
    """AutoFocus FocusPixel labels (gen_mask, data_workers.py:165-192): the oracle's restatement against masks produced by
    the reference's own anchor_worker with TRAIN.AUTO_FOCUS (tests/golden/make_focus_golden.py)."""
    import os
    from golden_util import anchor_case, ref_cfg
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'focus_mask_v1.npz'))
    lo, small, hi = [float(v) for v in g['thresholds']]
    cfg = ref_cfg()
    at = data_path.AnchorTarget(512, 16, cfg.network.ANCHOR_RATIOS, cfg.network.ANCHOR_SCALES, auto_focus=True, af_dc_low=lo,
                                af_dc_high=hi, af_small=small)
    n = int(g['count'])
    seen = set()
    for k in range(n):
        args, seed, _ = anchor_case(k)
        np.random.seed(seed)
        out = at(*args)
        assert np.array_equal(out[4], g['mask_%02d' % k]), k
        seen |= set(np.unique(out[4]).tolist())
    assert seen == {-1.0, 0.0, 1.0}
