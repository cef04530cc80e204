"""CPU: FocusChip generation (sniper_amd/chips_inference.py, restating lib/chips/chips_inference.py without OpenCV) on
hand-derived cases -- parity unpinned (no cv2 here to mint vectors), so these pin OUR documented semantics."""
import numpy as np

from sniper_amd.chips_inference import _bounding_rects, _dilate, gmask


def test_dilate_and_contour_rects():
    a = np.zeros((7, 7), np.uint8)
    a[3, 3] = 1
    assert _dilate(a, 3).sum() == 9 and _dilate(a, 3)[2:5, 2:5].all()
    assert sorted(zip(*_dilate(a, 2).nonzero())) == [(3, 3), (3, 4), (4, 3), (4, 4)]     # cv2 anchors a 2x2 kernel at (1, 1)
    ring = np.zeros((9, 9), np.uint8)
    ring[2:7, 2:7] = 255
    ring[4, 4] = 0
    # RETR_LIST: the component's outer border and the hole's border (the foreground ring around the hole)
    assert sorted(_bounding_rects(ring)) == [(2, 2, 5, 5), (3, 3, 3, 3)]
    two = np.zeros((6, 10), np.uint8)
    two[1, 1] = two[2, 2] = 255            # diagonal neighbours are one 8-connected component
    two[4, 8] = 255
    assert sorted(_bounding_rects(two)) == [(1, 1, 2, 2), (8, 4, 1, 1)]


def test_gmask_chips():
    m = np.zeros((30, 40), np.float32)            # a 480 x 640 crop at scale 1: 30 x 40 cells of 16 px
    m[5:8, 6:9] = 0.9
    m[20:25, 30:38] = 0.6
    chips = gmask(m, 3, 0.5, ms=4, im_width=640, im_height=480, cscale=1.0)
    # blob 1: cells x 5..9, y 4..8 after 3x3 dilation (5x5) -> >= 4 cells already; blob 2: x 29..38, y 19..25
    # (cv2.findContours reports the newest contour first: the lower blob, found later in the raster scan, leads)
    assert chips == [[464.0, 304.0, 624.0, 416.0], [80.0, 64.0, 160.0, 144.0]]
    # minimum chip side: a single hot cell grows to ms cells, clamped to the map, scaled back by cscale
    m = np.zeros((30, 40), np.float32)
    m[0, 39] = 1.0
    chips = gmask(m, 1, 0.5, ms=16, im_width=640, im_height=480, cscale=2.0)
    assert chips == [[(640 - 256) / 2.0, 0.0, 320.0, 128.0]]
    # nothing above the threshold: no chips
    assert gmask(np.zeros((10, 10), np.float32), 3, 0.5, ms=4, im_width=160, im_height=160) == []
    # two blobs whose minimum-size chips overlap merge into one chip on the second pass
    m = np.zeros((30, 40), np.float32)
    m[10, 10] = m[10, 14] = 1.0
    chips = gmask(m, 1, 0.5, ms=8, im_width=640, im_height=480, cscale=1.0)
    assert len(chips) == 1 and chips[0][0] <= 10 * 16 - 3 * 16 and chips[0][2] >= 15 * 16


def test_border_pruning_mask_equals_the_per_detection_rule():
    """Tester.get_detections' AutoFocus pruning (lib/inference.py:236-259 `check_valid`, applied per detection at :336-353):
    the vectorised mask keeps exactly the detections the per-detection rule keeps -- chips at every image border, boxes on,
    next to and far from each chip border."""
    from sniper_amd.inference import Tester
    rs = np.random.RandomState(4)
    W, H = 640, 480
    for chip in ([0, 0, 640, 480], [0.4, 0.6, 639.4, 479.6], [100, 50, 400, 300], [0, 120, 320, 480], [320, 0, 640.0, 240]):
        chip = np.asarray(chip, np.float64)
        base = rs.uniform(0, 1, (400, 4)) * [W, H, W, H]
        # snap a third of the coordinates onto / next to the chip borders (the rule's |d - c| < 10 boundary included)
        for k in range(4):
            sel = rs.rand(400) < 0.35
            base[sel, k] = chip[k] + rs.choice([-10.0, -9.999, -3, 0, 3, 9.999, 10.0, 10.001], sel.sum())
        dets = np.hstack((base, rs.rand(400, 1)))
        want = np.array([Tester._check_valid(d, chip, W, H) for d in dets], bool)
        got = Tester._valid_mask(dets, chip, W, H)
        assert got.dtype == bool and np.array_equal(got, want)
        if chip[0] < 0.5 and chip[1] < 0.5 and chip[2] >= W - 0.5 and chip[3] >= H - 0.5:
            assert want.all()                   # a chip that is the whole image prunes nothing
        else:
            assert 0 < want.sum() < len(want)
    assert Tester._valid_mask(np.zeros((0, 5)), [0, 0, 10, 10], W, H).shape == (0,)


def _loop_threshold(cscores, cboxes, thresh, num_classes):
    """lib/inference.py:289-295 as written there"""
    out = []
    for j in range(1, num_classes):
        inds = np.where(cscores[:, j] > thresh)[0]
        out.append(np.hstack((cboxes[inds, 0:4], cscores[inds, j, np.newaxis])))
    return out


def _loop_aggregate(scale_cls_dets, valid_ranges, num_images, num_classes):
    """lib/inference.py:170-190 as written there"""
    from sniper_amd.inference import _valid_range_filter
    problems = []
    for i in range(num_images):
        for j in range(1, num_classes):
            agg = [np.empty((0, 5), np.float32)]
            for all_cls_dets, vr in zip(scale_cls_dets, valid_ranges):
                for c in range(len(all_cls_dets[j][i])):
                    d = _valid_range_filter(np.asarray(all_cls_dets[j][i][c], np.float32).reshape(-1, 5), vr)
                    if d.shape[0] > 0:
                        agg.append(d)
            problems.append(np.vstack(agg))
    return problems


def test_vectorised_post_processing_equals_the_reference_loops():
    """Tester.get_detections / aggregate host work (score threshold per class, chip-border pruning, valid-range merge into the
    per (image, class) NMS problems): the array formulations return exactly what the reference's nested loops return -- same
    rows, same order, same dtypes -- including empty classes, empty chips and chips without any detection."""
    from sniper_amd.inference import Tester, aggregate_problems, prune_chip_border, threshold_detections
    rs = np.random.RandomState(11)
    NC, R = 9, 60
    for dtype in (np.float32, np.float64):
        cscores = rs.rand(R, NC).astype(np.float32) ** 6          # most scores under the threshold
        cscores[:, 3] = 0                                           # an empty class
        cboxes = (rs.rand(R, 4) * 300).astype(dtype)
        got, want = threshold_detections(cscores, cboxes, 0.05, NC), _loop_threshold(cscores, cboxes, 0.05, NC)
        assert len(got) == len(want) == NC - 1
        for g, w in zip(got, want):
            assert g.dtype == w.dtype and g.shape == w.shape and np.array_equal(g, w)
        assert got[2].shape == (0, 5) and sum(len(g) for g in got) > 0
    # pruning: per class float64 rows in image coordinates
    crop = np.array([100.0, 40.0, 420.0, 300.0])
    per_class = [np.hstack((np.sort(rs.rand(n, 4) * 320, axis=1)[:, [0, 1, 2, 3]], rs.rand(n, 1))).astype(np.float32)
                 for n in (0, 25, 3, 0, 40, 1, 0, 12)]
    per_class[1][:5, 0] = 4.0                                       # next to the chip's left border -> pruned
    got = prune_chip_border(per_class, crop, 640, 480)
    for a, g in zip(per_class, got):
        d = np.array(a, dtype=np.float64).reshape(-1, 5)
        d[:, 0] += crop[0]; d[:, 2] += crop[0]; d[:, 1] += crop[1]; d[:, 3] += crop[1]
        keep = np.array([Tester._check_valid(r, crop, 640, 480) for r in d], bool) if len(d) else np.zeros(0, bool)
        assert g.dtype == np.float64 and np.array_equal(g, d[keep].reshape(-1, 5))
    assert len(got[1]) <= 20
    # aggregation over 3 scales, 3 images with 1 / 2 / 0..3 chips
    n_img = 3
    scales = []
    for s_i in range(3):
        chips_per_image = [1, 2, s_i]
        scales.append([[[np.hstack((np.sort(rs.rand(n, 4) * 400, axis=1), rs.rand(n, 1))).astype(np.float64 if s_i else np.float32)
                         for n in rs.randint(0, 6, chips_per_image[i])] for i in range(n_img)] for _ in range(NC)])
    for sc in scales:                                              # a class without detections anywhere
        sc[4] = [[np.zeros((0, 5), np.float32) for _ in chips] for chips in sc[4]]
    vr = ((-1, 90), (32, 180), (75, -1))
    got, want = aggregate_problems(scales, vr, n_img, NC), _loop_aggregate(scales, vr, n_img, NC)
    assert len(got) == len(want) == n_img * (NC - 1)
    for g, w in zip(got, want):
        assert g.dtype == np.float32 and g.shape == w.shape and np.array_equal(g, w)
    assert sum(len(g) for g in got) > 0 and any(len(g) == 0 for g in got)
    # the stacked form (one array + rows per problem: what the batched soft-NMS uploads) holds the same rows
    rows, sizes = aggregate_problems(scales, vr, n_img, NC, stacked=True)
    assert rows.dtype == np.float32 and np.array_equal(sizes, [len(w) for w in want])
    assert np.array_equal(rows, np.concatenate([w.reshape(-1, 5) for w in want]))
    # ... and so does the path that reads the chips' rows as the GPU returns them (grouped by class + rows per class)
    from sniper_amd.inference import _Detections
    marked = []
    for sc in scales:
        d = _Detections(sc)
        for i in range(n_img):
            for c in range(len(sc[1][i])):
                per = [np.asarray(sc[j][i][c], np.float64).reshape(-1, 5) for j in range(1, NC)]
                d.compact[(i, c)] = (np.concatenate(per), np.array([len(a) for a in per]))
        marked.append(d)
    rows2, sizes2 = aggregate_problems(marked, vr, n_img, NC, stacked=True)
    assert np.array_equal(rows2, rows) and np.array_equal(sizes2, sizes)
    assert aggregate_problems([], (), 2, NC, stacked=True)[1].tolist() == [0] * (2 * (NC - 1))


def test_reference_aggregation_baseline_matches_the_vectorised_one():
    """bench.py's inference cpu_baseline (oracle/inference_ref.py: Tester.aggregate's loops as the reference writes them,
    lib/inference.py:166-190) builds the same (image, class) NMS problems, row for row, as sniper_amd.inference.aggregate_problems;
    its soft-NMS (the reference's compiled cpu_nms.pyx where built, else the C restatement) keeps what the oracle's keeps."""
    import oracle
    from oracle import inference_ref
    from sniper_amd.inference import aggregate_problems
    rs = np.random.RandomState(3)
    num_images, num_classes = 3, 6
    vr = ((75, -1), (32, 180), (-1, 75))
    scale_dets = []
    for s in range(3):
        allc = [[[] for _ in range(num_images)] for _ in range(num_classes)]
        for i in range(num_images):
            n_chips = 1 + (s * (i + 1)) % 3
            for j in range(1, num_classes):
                allc[j][i] = []
                for c in range(n_chips):
                    n = int(rs.randint(0, 9))
                    xy = rs.uniform(0, 400, (n, 2))
                    wh = rs.uniform(5, 260, (n, 2))
                    allc[j][i].append(np.hstack((xy, xy + wh, rs.uniform(0.01, 1, (n, 1)))).astype(np.float32) if n else [])
        scale_dets.append(allc)
    a = inference_ref.aggregate_problems(scale_dets, vr, num_images, num_classes)
    b = aggregate_problems(scale_dets, vr, num_images, num_classes)
    assert len(a) == len(b) == num_images * (num_classes - 1)
    for u, v in zip(a, b):
        assert u.shape == v.shape and np.array_equal(u, v)
    big = max(a, key=len)
    assert len(big) > 3
    assert np.array_equal(inference_ref.nms_problem(big, 0.55), oracle.soft_nms(big, sigma=0.55, Nt=0.3, threshold=0.001, method=2))
    assert inference_ref.kind() in ('reference', 'port')


def test_bench_focus_maps_give_the_specified_c5_workload():
    """bench.py's injected FocusPixel maps (~10 % positive pixels in blobs) through the real FocusChip generation:
    the positive fraction of every map, the chips per image at the two finer scales (what BENCH prints as
    chips_per_image_by_scale) and the share of the finest scale's pixels that is actually processed."""
    from bench import focus_map_blobs
    from sniper_amd import config as cfgmod
    from sniper_amd.chips_inference import add_chips
    from sniper_amd.data.im_worker import target_scale
    cfg = cfgmod.res101_e2e_autofocus()
    roidb = [{'width': 640, 'height': 480, 'inference_crops': np.array([[0, 0, 640, 480]])} for _ in range(8)]
    chips, areas = [], []
    for s in range(2):
        maps = []
        for i, r in enumerate(roidb):
            sc = target_scale(640, 480, cfg.TEST.SCALES[s])
            per = []
            for j, c in enumerate(r['inference_crops']):
                h, w = int(np.ceil((c[3] - c[1]) * sc / 16)), int(np.ceil((c[2] - c[0]) * sc / 16))
                m = focus_map_blobs(s, i, j, np.zeros((2, h, w), np.float32))
                assert m.shape == (2, h, w) and np.allclose(m.sum(0), 1.0)
                assert 0.07 <= float((m[1] > 0.5).mean()) <= 0.13
                assert np.array_equal(m, focus_map_blobs(s, i, j, np.zeros((2, h, w), np.float32)))      # deterministic
                per.append(m)
            maps.append(per)
        areas.append(add_chips(roidb, maps, s, cfg))
        chips.append([len(r['inference_crops']) for r in roidb])
    assert chips[0] == [1] * 8 and all(1 <= n <= 4 for n in chips[1]) and 12 <= sum(chips[1]) <= 24
    assert sum(chips[1]) == 17                                      # the line profiles/r03_bench_v2.json reports
    assert 0.2 <= areas[1][0] / areas[1][1] <= 0.5                  # a third of the finest scale's pixels is run


def test_native_focus_chips_equal_the_python_statement():
    """sn_focus_chips_host (what gmask calls) against gmask_reference (the scipy.ndimage statement of lib/chips/chips_inference.py
    :12-89) on random maps: blobs, rings and frames (holes), noise, empty maps; every dilation size, threshold and minimum side of
    the configs and a few more; crops that end inside the last cell.  Same chips, same float64 values; the order is cv2's."""
    from sniper_amd.chips_inference import gmask, gmask_reference
    rs = np.random.RandomState(4)
    n_chips = 0
    for trial in range(500):
        H, W = int(rs.randint(3, 90)), int(rs.randint(3, 126))
        m = np.full((H, W), 0.01, np.float32)
        yy, xx = np.mgrid[0:H, 0:W]
        kind = trial % 5
        if kind == 0:
            for _ in range(rs.randint(1, 6)):
                cy, cx, r = rs.randint(0, H), rs.randint(0, W), rs.randint(1, 9)
                m[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 0.9
        elif kind == 1:
            for _ in range(rs.randint(1, 4)):
                cy, cx, r = rs.randint(0, H), rs.randint(0, W), rs.randint(3, 14)
                d2 = (yy - cy) ** 2 + (xx - cx) ** 2
                m[(d2 <= r * r) & (d2 >= (r - 2) ** 2)] = 0.9
        elif kind == 2:
            m = rs.rand(H, W).astype(np.float32) ** int(rs.randint(1, 8))
        elif kind == 3:
            for _ in range(rs.randint(1, 4)):
                y0, x0 = rs.randint(0, max(1, H - 2)), rs.randint(0, max(1, W - 2))
                y1, x1 = rs.randint(y0, H), rs.randint(x0, W)
                m[y0:y1 + 1, x0:x1 + 1] = 0.9
                if y1 - y0 > 3 and x1 - x0 > 3:
                    m[y0 + 2:y1 - 1, x0 + 2:x1 - 1] = 0.0
        d, thr, ms = int(rs.choice([1, 2, 3, 4, 5])), float(rs.choice([0.02, 0.2, 0.5])), int(rs.choice([1, 4, 16, 20]))
        cs = float(rs.choice([0.8, 1.6667, 2.9166]))
        imw, imh = W * 16 - int(rs.randint(0, 16)), H * 16 - int(rs.randint(0, 16))
        got, want = gmask(m, d, thr, ms, imw, imh, cs), gmask_reference(m, d, thr, ms, imw, imh, cs)
        # (round 5: the native code follows borders and reports the chips in cv2's order -- newest contour first; the scipy statement
        # numbers components in raster order, holes last: the same chips, compared as sets here and WITH their order against the
        # restated cv2 calls of oracle/cv_contours.py)
        assert sorted(map(tuple, got)) == sorted(tuple(float(v) for v in c) for c in want), (trial, kind, (H, W), d, thr, ms)
        if trial % 4 == 0:
            from oracle import cv_contours
            assert got == cv_contours.gmask(m, d, thr, ms, imw, imh, cs), (trial, kind, (H, W), d, thr, ms)
        n_chips += len(got)
    assert n_chips > 500


def test_native_aggregation_equals_the_numpy_statement():
    """sn_aggregate_problems_host (one pass over the chips' rows as the GPU returns them) against the numpy statement of the
    valid-range merge (itself held against the reference's loops above): random row counts incl. empty chips and classes, three
    scales with 1 / 1 / 1-3 chips per image, open and two-sided valid ranges.  Same rows, same order, same float32 bits."""
    from sniper_amd.inference import _Detections, _aggregate_stacked_native, aggregate_problems
    rs = np.random.RandomState(9)
    NC, n_img = 13, 5
    nc = NC - 1

    def scale(chips_per_image):
        d = _Detections([[[None] * chips_per_image[i] for i in range(n_img)] for _ in range(NC)])
        for i in range(n_img):
            for c in range(chips_per_image[i]):
                lens = rs.multinomial(int(rs.randint(0, 400)), np.ones(nc) / nc).astype(np.int64)
                if rs.rand() < 0.2:
                    lens[:] = 0
                n = int(lens.sum())
                xy, wh = rs.uniform(0, 600, (n, 2)), rs.uniform(1, 300, (n, 2))
                big = np.hstack((xy, xy + wh, rs.uniform(0.001, 1, (n, 1)))).astype(np.float64)
                ends = np.cumsum(lens)
                for j in range(nc):
                    d[j + 1][i][c] = big[ends[j] - lens[j]:ends[j]]
                d.compact[(i, c)] = (big, lens)
        return d
    scales = [scale([1] * n_img), scale([1] * n_img), scale([1, 2, 3, 0, 2])]
    for vr in (((-1, -1),) * 3, ((-1, 90), (32, 180), (75, -1)), ((40, -1), (-1, 60), (10, 400))):
        got = _aggregate_stacked_native(scales, vr, n_img, nc)
        assert got is not None
        want_rows, want_sizes = aggregate_problems([list(s) for s in scales], vr, n_img, NC, stacked=True)     # (no `compact`: numpy)
        assert got[0].dtype == np.float32 and np.array_equal(got[0], want_rows) and np.array_equal(got[1], want_sizes)
        assert len(want_rows) > 100
    # a chip without the compact form sends the whole call to the numpy statement
    del scales[1].compact[(2, 0)]
    assert _aggregate_stacked_native(scales, ((-1, -1),) * 3, n_img, nc) is None


def test_cap_detections_per_image_equals_the_per_class_loops():
    """inference.cap_detections_per_image against the reference's statement of the MAX_PER_IMAGE rule (lib/inference.py:203-211):
    hstack the scores, threshold = the max_per_image-th best, per class `where(score >= threshold)`; ties, empty classes, fewer
    rows than the cap"""
    from sniper_amd.inference import cap_detections_per_image
    rs = np.random.RandomState(4)
    for case in range(40):
        nc = int(rs.randint(1, 81))
        per_class = []
        for j in range(nc):
            n = int(rs.randint(0, 9)) if rs.rand() < 0.8 else 0
            d = rs.rand(n, 5).astype(np.float32)
            if case % 3 == 0 and n:
                d[:, 4] = np.round(d[:, 4] * 4) / 4           # many equal scores: ties at the threshold stay
            per_class.append(d)
        cap = int(rs.randint(1, 120))
        scores = np.hstack([d[:, -1] for d in per_class]) if per_class else np.zeros(0)
        got = cap_detections_per_image(per_class, cap)
        if len(scores) <= cap:
            assert got is None
            continue
        thresh = np.sort(scores)[-cap]
        want = [d[np.where(d[:, -1] >= thresh)[0], :] for d in per_class]
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b)
