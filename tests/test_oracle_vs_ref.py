"""Pin the oracle: the C/numpy restatements must reproduce the reference's own code (run in this
container through oracle/ref_py.py + oracle/_ref) on seeded synthetic inputs.  Skipped where the
reference checkout is absent (GPU box); there the committed golden vectors take over."""
import copy

import numpy as np
import pytest

import oracle
from oracle import data_path, ref_py
from sniper_amd import config as cfgmod
from sniper_amd.synthetic import make_roidb

pytestmark = pytest.mark.ref


@pytest.fixture(scope="module")
def ref():
    return ref_py.load()


def _ref_cfg():
    c = cfgmod.res101_e2e()
    return c


def test_chips_generate_matches_cchips(ref):
    rs = np.random.RandomState(1)
    n_checked = 0
    for t in range(200):
        W, H = int(rs.randint(300, 2100)), int(rs.randint(300, 1600))
        n = int(rs.randint(0, 40))
        side = np.exp(rs.uniform(np.log(4), np.log(400), size=n))
        x1 = rs.uniform(0, W - 2, size=n)
        y1 = rs.uniform(0, H - 2, size=n)
        b = np.stack((x1, y1, np.minimum(x1 + side, W - 2), np.minimum(y1 + side, H - 2)), 1).astype(np.float32)
        stride = int(rs.randint(56, 60))
        ref_py.srand(1000 + t)
        want = np.array(ref.chips.generate(np.ascontiguousarray(b), W, H, 512, stride), np.float32).reshape(-1, 4)
        ncand = oracle.candidate_chips(W, H, 512, stride).shape[0]
        perm = oracle.shuffle_perm(ncand, 1000 + t)
        got = oracle.chips_generate(b, W, H, 512, stride, perm)
        assert got.shape == want.shape and np.array_equal(got, want), (t, W, H, n)
        n_checked += len(want)
    assert n_checked > 200


def test_iou_matches_bbox_pyx(ref):
    rs = np.random.RandomState(2)
    for t in range(20):
        a = rs.uniform(0, 500, size=(rs.randint(1, 200), 4))
        a[:, 2:] += a[:, :2]
        q = np.round(rs.uniform(0, 500, size=(rs.randint(1, 50), 4)))
        q[:, 2:] += q[:, :2]
        assert np.array_equal(oracle.bbox_overlaps(a, q), ref.bbox.bbox_overlaps_cython(a, q))
        assert np.array_equal(oracle.ignore_overlaps(a, q), ref.bbox.ignore_overlaps_cython(a, q))
    # exact containment / identical boxes give exactly 1.0
    b = np.array([[10., 10., 50., 60.]])
    assert oracle.bbox_overlaps(b, b)[0, 0] == 1.0


def test_anchors_match_generate_anchor(ref):
    for ratios, scales, stride in (((0.5, 1, 2), (2, 4, 7, 10, 13, 16, 24), 16), ((0.5, 1, 2), (1, 2, 4, 8, 12), 32)):
        want = ref.generate_anchor.generate_anchors(base_size=stride, ratios=list(ratios),
                                                    scales=list(np.array(scales, np.float32)))
        got = data_path.generate_anchors(stride, ratios, np.array(scales, np.float32))
        assert np.array_equal(want, got)


def _mk_workers(ref, cfg, stride):
    np.random.seed(0)
    cw = ref.data_workers.chip_worker(cfg, 512)
    cw.chip_stride = stride
    cw.chip_generator = ref.chip_generator.chip_generator(chip_stride=stride, use_cpp=True)
    aw = ref.data_workers.anchor_worker(cfg, 512)
    return cw, aw


def test_chip_extractor_box_assigner_anchor_worker(ref):
    cfg = _ref_cfg()
    stride = 56
    cw, aw = _mk_workers(ref, cfg, stride)
    roidb = make_roidb(40, seed=3, n_proposals=300)
    at = data_path.AnchorTarget(512, 16, cfg.network.ANCHOR_RATIOS, cfg.network.ANCHOR_SCALES)
    assert np.array_equal(at.anchors, aw.all_anchors)
    n_chips = 0
    for i, r in enumerate(roidb):
        # --- chip extraction: same libc seed per (image, scale) call sequence
        ref_py.srand(7000 + i)
        want = cw.chip_extractor(copy.deepcopy(r))
        seeds = iter([7000 + i])

        state = {"first": True}

        def perm_fn(n, _s=state, _i=i):
            # the reference draws from one continuing libc stream per image: seed once, continue after
            p = oracle.shuffle_perm(n, 7000 + _i if _s["first"] else -1)
            _s["first"] = False
            return p

        got = data_path.chip_extractor(r, cfg.TRAIN.SCALES, cfg.TRAIN.VALID_RANGES, 512, stride, perm_fn)
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert np.array_equal(g[0], w[0]) and g[1:] == w[1:]
        # --- box assignment (incl. negative chip mining)
        r1 = copy.deepcopy(r)
        r1['crops'] = want
        ref_py.srand(9000 + i)
        wp, wn, wnp = cw.box_assigner(r1)
        state["first"] = True

        def perm_fn2(n, _s=state, _i=i):
            p = oracle.shuffle_perm(n, 9000 + _i if _s["first"] else -1)
            _s["first"] = False
            return p

        r2 = copy.deepcopy(r)
        r2['crops'] = got
        gp, gn, gnp = data_path.box_assigner(r2, cfg.TRAIN.SCALES, cfg.TRAIN.VALID_RANGES, 512, stride, True, perm_fn2)
        assert len(gp) == len(wp) and all(np.array_equal(a, b) for a, b in zip(gp, wp))
        assert len(gn) == len(wn)
        for g, w in zip(gn, wn):
            assert np.array_equal(g[0], w[0]) and g[1:] == w[1:]
        assert all(np.array_equal(a, b) for a, b in zip(gnp, wnp))
        # --- anchor labelling for every chip of this image
        gtids = np.where(r['max_overlaps'] == 1)[0]
        for ci, crop in enumerate(want):
            args = lambda: [[512, 512, crop[1]], crop[0].copy(), crop[1], wp[ci], gtids, r['boxes'][gtids].copy(),
                            r['boxes'].copy(), r['max_classes'][gtids].reshape(-1, 1)]
            np.random.seed(100 + ci)
            w = aw.worker(args())
            np.random.seed(100 + ci)
            g = at(*args())
            wl = np.asarray(w[0], np.float32).reshape(-1)
            assert np.array_equal(g[0], wl)
            pids = tuple(np.asarray(p).astype(np.int64) for p in w[2])
            dense_t = np.zeros((84, 32, 32), np.float32)
            dense_w = np.zeros((84, 32, 32), np.float32)
            if len(pids[0]) > 0:
                dense_t[pids] = w[1]
                dense_w[pids] = 1.0
            assert np.array_equal(g[1], dense_t) and np.array_equal(g[2], dense_w)
            assert np.array_equal(g[3], np.asarray(w[3], np.float32))
            n_chips += 1
    assert n_chips > 100


def test_soft_and_hard_nms_match_cpu_nms_pyx():
    """oracle soft-NMS / cpu_nms restatements against the reference's own lib/nms/cpu_nms.pyx, compiled by oracle/build.py
    (cpu_soft_nms verbatim).  Found by this test: Cython promotes the `+ 1` beside C floats to a double literal, so `ua` is a
    double sum rounded once -- float arithmetic is off by one ulp in ~40 % of the decayed scores."""
    import sys
    sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), 'golden'))
    from make_nms_golden import load_ref_cpu_nms, problem
    refnms = load_ref_cpu_nms()
    rs = np.random.RandomState(5)
    checked = 0
    for n in (1, 3, 25, 120, 500):
        for method in (0, 1, 2):
            for thr, quant in ((0.001, None), (0.02, 25)):
                d = problem(rs, n, quant)
                want = np.asarray(refnms.cpu_soft_nms(d.copy(), 0.55, 0.3, thr, method), np.float32)
                got = oracle.soft_nms(d.copy(), 0.55, 0.3, thr, method)
                assert got.shape == want.shape and np.array_equal(got, want), (n, method, thr)
                checked += 1
        for thr in (0.3, 0.5, 0.7):
            d = problem(rs, n, 40)
            assert list(oracle.cpu_nms(d, thr)) == list(refnms.cpu_nms(d.copy(), thr)), (n, thr)
    assert checked == 30


def test_hard_nms_matches_reference_numpy_nms():
    """oracle.nms_sorted (the nms_kernel.cu rule: suppress IoU > thresh on score-sorted boxes) against the reference's own
    lib/nms/nms.py:90-127 `nms` (keeps ovr <= thresh) and its `nms_wrapper` / `soft_nms` front ends, imported unchanged
    with the compiled cpu_nms module of oracle/_ref under their import names."""
    import importlib.util
    import sys
    import types
    sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), 'golden'))
    from make_nms_golden import load_ref_cpu_nms, problem
    saved = {k: sys.modules.get(k) for k in ('cpu_nms', 'gpu_nms')}
    try:
        sys.modules['cpu_nms'] = load_ref_cpu_nms()
        g = types.ModuleType('gpu_nms')
        g.gpu_nms = None                       # nvcc-only; nms.py merely imports the name (lib/nms/nms.py:3-4)
        sys.modules['gpu_nms'] = g
        spec = importlib.util.spec_from_file_location('ref_nms_py', '/root/reference/lib/nms/nms.py')
        refnms = importlib.util.module_from_spec(spec)
        sys.dont_write_bytecode = True
        spec.loader.exec_module(refnms)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    rs = np.random.RandomState(13)
    for n in (1, 6, 80, 400, 1500):
        for thr in (0.3, 0.5, 0.7):
            d = problem(rs, n, 30 if n % 2 == 0 else None)
            want = [int(i) for i in refnms.nms(d, thr)]
            order = d[:, 4].argsort()[::-1]
            got = order[oracle.nms_sorted(d[order], thr)]
            assert [int(i) for i in got] == want, (n, thr)
            assert [int(i) for i in refnms.nms_wrapper(thr, -1).process(d)] == want
    d = problem(rs, 200, 20)
    soft = refnms.nms_wrapper(-1, 0.55).process(d.copy())
    assert np.array_equal(np.asarray(soft, np.float32), oracle.soft_nms(d.copy(), 0.55, 0.3, 0.001, 2))


def test_box_coders_match_bbox_transform(ref):
    """oracle data_path.{bbox_transform, bbox_pred, clip_boxes, filter_boxes} against the reference's
    lib/bbox/bbox_transform.py (nonlinear_transform :64-90, nonlinear_pred :93-130, clip_boxes :35-50, filter_boxes :52-61):
    bit-equal on float32 and float64 inputs, class-agnostic (4 columns) and per-class (4*K columns) deltas, empty inputs."""
    bt = ref.bbox_transform
    rs = np.random.RandomState(21)
    for dt in (np.float32, np.float64):
        for n, k in ((0, 1), (1, 1), (300, 1), (57, 3)):
            c = rs.uniform(0, 500, (n, 2))
            wh = rs.uniform(1, 200, (n, 2))
            boxes = np.hstack((c - wh / 2, c + wh / 2)).astype(dt)
            c2 = c + rs.normal(0, 10, (n, 2))
            wh2 = wh * np.exp(rs.normal(0, 0.3, (n, 2)))
            gt = np.hstack((c2 - wh2 / 2, c2 + wh2 / 2)).astype(dt)
            deltas = (rs.standard_normal((n, 4 * k)) * 0.4).astype(dt)
            if n:
                assert np.array_equal(data_path.bbox_transform(boxes.copy(), gt.copy()), bt.bbox_transform(boxes.copy(), gt.copy()))
            got, want = data_path.bbox_pred(boxes.copy(), deltas.copy()), bt.bbox_pred(boxes.copy(), deltas.copy())
            assert got.shape == want.shape and got.dtype == want.dtype and np.array_equal(got, want), (dt, n, k)
            if n:
                assert np.array_equal(data_path.clip_boxes(got.copy(), (384, 511)), bt.clip_boxes(want.copy(), (384, 511)))
                assert np.array_equal(data_path.filter_boxes(boxes, 12), bt.filter_boxes(boxes, 12))
