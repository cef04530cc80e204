"""-m gpu: the test-time path: GPU image preparation, test iterators, box decoding,
Tester.detect / get_detections / aggregate with batched soft-NMS, FocusChip generation, on synthetic images."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import dev  # noqa: E402


class _Imdb(object):
    def __init__(self, n):
        self.num_classes, self.classes, self.name, self.result_path = n, ['c%d' % i for i in range(n)], 'synthetic', None


def _roidb(n, rs, sizes=((480, 640), (640, 480))):
    out = []
    for i in range(n):
        h, w = sizes[i % len(sizes)] if i >= n // 2 else sizes[0]
        out.append({'image': rs.randint(0, 256, (h, w, 3)).astype(np.uint8), 'width': w, 'height': h, 'flipped': False,
                    'gt_overlaps': np.zeros((1, 81), np.float32)})
    return out


def test_im_prepare_is_bit_exact_with_opencv_8bit_inter_linear():
    """sn_im_prepare == im_worker.worker / worker_autofocus (lib/data_utils/data_workers.py:49-121) with cv2.resize(INTER_LINEAR) on
    uint8 restated from OpenCV's published fixed-point algorithm (oracle/cv_resize.py, cross-checked against its scalar C twin in
    tests/test_oracle_cv_resize.py): flips, crops beyond the image, up- and down-scaling, the 2 x 2 INTER_AREA substitution with
    odd sizes, the SNIPER train / test scales, tiny crops, padding smaller and larger than the resized image.  Byte work: equal."""
    import ctypes
    from oracle import cv_resize
    from sniper_amd import hip
    rs = np.random.RandomState(0)
    means = np.array([103.939, 116.779, 123.68], np.float64)
    cases = 0
    for (H, W) in ((37, 53), (48, 64), (31, 31), (5, 7)):
        im = rs.randint(0, 256, (H, W, 3)).astype(np.uint8)
        d = torch.from_numpy(im).to(dev())
        for scale in (1.0, 1.7, 0.5, 3.0, 2.25, 512.0 / 480.0, 1.0 / 0.6, 2.9167, 0.8, 1400.0 / 480.0, 0.3, 2.0, 0.5000001, 0.25):
            for flip in (0, 1):
                x1, y1 = int(rs.randint(-3, W // 2)), int(rs.randint(-3, H // 2))
                x2, y2 = int(rs.randint(max(x1, 0) + 1, W + 4)), int(rs.randint(max(y1, 0) + 1, H + 4))
                crop = (0, 0, W, H) if rs.rand() < 0.3 else (x1, y1, x2, y2)
                cw, ch = min(crop[2], W) - max(crop[0], 0), min(crop[3], H) - max(crop[1], 0)
                rh, rw = cv_resize.dsize_of(ch, cw, scale, scale)
                if rh < 1 or rw < 1:
                    continue
                out_hw = (int(rs.randint(max(rh - 6, 1), rh + 9)), int(rs.randint(max(rw - 6, 1), rw + 9)))
                out = torch.full((3,) + out_hw, 9.0, dtype=torch.float32, device=dev())
                hw = (ctypes.c_int32 * 2)()
                hip.call('sn_im_prepare', d, H, W, crop[0], crop[1], crop[2], crop[3], float(scale), flip,
                         means.ctypes.data_as(ctypes.c_void_p), out, out_hw[0], out_hw[1], hw, hip.stream())
                want, got_hw = cv_resize.im_prepare(im, crop, scale, bool(flip), means, out_hw)
                assert (hw[0], hw[1]) == got_hw == (rh, rw)
                got = out.cpu().numpy()
                assert np.array_equal(got, want), (H, W, crop, scale, flip, out_hw, np.abs(got - want).max(), (got != want).mean())
                cases += 1
    assert cases > 100
    # a 640 x 480 image at the three AutoFocus test scales, whole image (what the first scale of a pass prepares)
    im = rs.randint(0, 256, (480, 640, 3)).astype(np.uint8)
    d = torch.from_numpy(im).to(dev())
    for scale in (1.0, 800.0 / 480.0, 1400.0 / 480.0):
        rh, rw = cv_resize.dsize_of(480, 640, scale, scale)
        out = torch.empty((3, rh + 3, rw + 5), dtype=torch.float32, device=dev())
        hip.call('sn_im_prepare', d, 480, 640, 0, 0, 640, 480, float(scale), 0, means.ctypes.data_as(ctypes.c_void_p), out, rh + 3, rw + 5,
                 None, hip.stream())
        want, _ = cv_resize.im_prepare(im, (0, 0, 640, 480), scale, False, means, (rh + 3, rw + 5))
        assert np.array_equal(out.cpu().numpy(), want)


def test_bbox_decode_matches_numpy_bbox_pred():
    """sn_bbox_decode == bbox_pred + clip_boxes + /scale of lib/inference.py:127-131, in float64 like numpy."""
    from oracle import data_path
    from sniper_amd import hip
    rs = np.random.RandomState(1)
    B, R = 3, 50
    rois = np.zeros((B * R, 5), np.float32)
    rois[:, 0] = np.repeat(np.arange(B), R)
    c = rs.uniform(0, 500, (B * R, 2))
    wh = rs.uniform(2, 300, (B * R, 2))
    rois[:, 1:3], rois[:, 3:5] = c - wh / 2, c + wh / 2
    deltas = (rs.standard_normal((B, R, 4)) * 0.5).astype(np.float32)
    info = np.array([[480, 640, 1.5], [512, 512, 0.75], [300, 400, 2.0]], np.float32)
    out = torch.empty((B, R, 4), dtype=torch.float64, device=dev())
    td = lambda a: torch.from_numpy(a).to(dev())
    hip.call('sn_bbox_decode', td(rois), td(deltas), td(info), out, B, R, hip.stream())
    got = out.cpu().numpy()
    for b in range(B):
        want = data_path.bbox_pred(rois[b * R:(b + 1) * R, 1:], deltas[b])
        want = data_path.clip_boxes(want, info[b, :2]) / info[b, 2]
        # float64 arithmetic; np.exp on the float32 deltas is a float32 routine whose last bit is implementation defined
        # (the device narrows a double exp): one float32 ulp of exp(dw) on a <= 2000-pixel box
        assert np.allclose(got[b], want, rtol=1e-6, atol=2e-4), np.abs(got[b] - want).max()


def test_autofocus_pipeline_end_to_end():
    """Coarse-to-fine inference on synthetic images with the R101 AutoFocus test graph at reduced scales: the three
    iterators, Module re-binding per batch shape, Tester.detect on the device decode, FocusChips, multi-scale aggregation."""
    import sniper_amd.mx as mx
    from sniper_amd import config as cfgmod
    from sniper_amd.inference import Tester, imdb_detection_wrapper, nms_worker
    from sniper_amd.iterators.MNIteratorTest import MNIteratorTest
    from sniper_amd.iterators.MNIteratorTestAutoFocus import MNIteratorTestAutoFocus
    from sniper_amd.symbols.faster import resnet_mx_101_e2e as rn
    rs = np.random.RandomState(3)
    roidb = _roidb(4, rs, sizes=((240, 320),))
    cfg = cfgmod.res101_e2e_autofocus()
    cfg.TEST.SCALES = ((120, 160), (240, 320))
    cfg.TEST.BATCH_IMAGES = (2, 2)
    cfg.TEST.VALID_RANGES = ((40, -1), (-1, 60))
    cfg.TEST.DO_PRUNING = (False, True)
    cfg.TEST.CHIP_HYPERPARAMS = ((3, 0.3, 4), (-1, -1, -1))
    cfg.TEST.MAX_PER_IMAGE = 50
    cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N = 500, 100
    # iterator contracts
    it = MNIteratorTest(roidb, cfg, test_scale=(120, 160), batch_size=2, crop_size=None)
    assert [k for k, _ in it.provide_data] == ['data', 'im_info', 'im_ids'] and dict(it.provide_data)['data'] == (2, 3, 120, 160)
    b = it.next()
    info = b.data[1].asnumpy()
    assert np.allclose(info[:, 2], 0.5) and np.array_equal(info[:, :2], [[120, 160]] * 2)
    for r in roidb:
        r['inference_crops'] = np.array([[0, 0, r['width'], r['height']]])
    it2 = MNIteratorTestAutoFocus(roidb, cfg, test_scale=(240, 320), batch_size=2, crop_size=None)
    assert [k for k, _ in it2.provide_data] == ['data', 'im_info', 'im_ids', 'chip_ids'] and len(it2) == 4
    # the whole coarse-to-fine wrapper with random weights (detections are noise; structure and invariants are checked)
    net = rn.resnet_mx_101_e2e
    all_boxes = imdb_detection_wrapper(net, cfg, _Imdb(81), roidb, [mx.gpu(0)], None, None)
    assert len(all_boxes) == 81 and len(all_boxes[1]) == 4
    for i in range(4):
        sc = np.sort(np.hstack([all_boxes[j][i][:, 4] for j in range(1, 81)]))[::-1]
        # MAX_PER_IMAGE (:203-209): everything kept scores at least the 50th best (ties at that score all stay --
        # random weights make nearly every score tie)
        assert len(sc) <= 50 or sc[-1] >= sc[49]
        for j in range(1, 81):
            d = all_boxes[j][i]
            assert d.shape[1] == 5 and np.isfinite(d).all()
            if d.shape[0]:
                assert d[:, 0].min() >= -1e-6 and d[:, 2].max() <= roidb[i]['width'] + 1e-3
                assert (np.diff(d[:, 4]) <= 1e-7).all()            # soft-NMS emits rows by descending decayed score
    for r in roidb:                                 # FocusChips of the second scale lie inside their image
        c = np.asarray(r['inference_crops'], np.float64).reshape(-1, 4)
        if c.shape[0]:
            assert c[:, 0].min() >= 0 and c[:, 1].min() >= 0 and c[:, 2].max() <= r['width'] + 1e-6 and c[:, 3].max() <= r['height'] + 1e-6
    # batched soft-NMS worker == one problem at a time
    w = nms_worker(-1, 0.55)
    probs = [np.hstack((rs.uniform(0, 50, (n, 2)), rs.uniform(60, 120, (n, 2)), rs.uniform(0.01, 1, (n, 1)))).astype(np.float32)
             for n in (5, 40, 0, 17)]
    many = w.worker_many([p.copy() for p in probs])
    for p, m in zip(probs, many):
        assert np.array_equal(m, w.worker(p.copy()))


def test_det_compact_equals_the_host_threshold_and_prune():
    """sn_det_compact == threshold_detections (lib/inference.py:289-295) followed by prune_chip_border (:336-353) for every
    chip of a batch, bit for bit: rows grouped by class, RoIs ascending, float64; with and without pruning; a chip with no
    surviving row, a chip whose every row survives, chips on the image border (those sides never prune)."""
    from sniper_amd import hip
    from sniper_amd.inference import threshold_detections, prune_chip_border
    rs = np.random.RandomState(11)
    B, R, NC = 5, 150, 21
    scores = rs.dirichlet(np.ones(NC) * 0.08, (B, R)).astype(np.float32)
    scores[1] = 0.0                                   # nothing passes
    scores[2] = 0.5                                   # everything passes
    boxes = rs.uniform(0, 200, (B, R, 4))
    boxes[..., 2:] += boxes[..., :2]
    boxes[3, :40, 0] = rs.uniform(0, 12, 40)          # rows hugging the left / top chip border
    boxes[4, :40, 1] = rs.uniform(0, 12, 40)
    crops = np.array([[30., 20., 430., 420.], [0., 0., 400., 400.], [100., 50., 500., 450.], [64., 0., 464., 400.],
                      [0., 48., 400., 448.]])
    wh = np.array([[640., 480.], [400., 400.], [500., 450.], [640., 400.], [400., 640.]])
    td = lambda z: torch.from_numpy(np.ascontiguousarray(z)).to(dev())
    for prune in (False, True):
        rows = torch.full((B, (NC - 1) * R, 5), float('nan'), dtype=torch.float64, device=dev())
        counts = torch.zeros((B, NC - 1), dtype=torch.int32, device=dev())
        hip.call('sn_det_compact', td(scores), td(boxes), crops if prune else None, wh if prune else None, 1e-3, 10.0, B, R, NC,
                 rows, counts, hip.stream())
        rows, counts = rows.cpu().numpy(), counts.cpu().numpy()
        for b in range(B):
            want = threshold_detections(scores[b], boxes[b], 1e-3, NC)
            if prune:
                want = prune_chip_border(want, crops[b], wh[b][0], wh[b][1])
            assert np.array_equal(counts[b], [len(w) for w in want]), (prune, b)
            stacked = np.concatenate([np.asarray(w, np.float64).reshape(-1, 5) for w in want])
            assert np.array_equal(rows[b, :len(stacked)], stacked), (prune, b)
        assert counts[1].sum() == 0 and (prune or counts[2].sum() == (NC - 1) * R)


def test_lanes_and_device_compaction_equal_the_host_loops():
    """The coarse-to-fine wrapper with three lanes (forwards of consecutive batches on their own streams) and the GPU threshold /
    prune returns exactly the per-scale detections and final boxes of one lane with the numpy loops -- on the eager first pass,
    the capturing second pass and a replayed third pass."""
    import copy
    import sniper_amd.mx as mx
    from sniper_amd import config as cfgmod
    from sniper_amd import inference
    from sniper_amd.symbols.faster import resnet_mx_101_e2e as rn
    rs = np.random.RandomState(5)
    base = _roidb(6, rs, sizes=((240, 320),))
    cfg = cfgmod.res101_e2e_autofocus()
    cfg.TEST.SCALES = ((240, 320), (480, 640))
    cfg.TEST.BATCH_IMAGES = (2, 2)
    cfg.TEST.VALID_RANGES = ((40, -1), (-1, 60))
    cfg.TEST.DO_PRUNING = (False, True)
    cfg.TEST.CHIP_HYPERPARAMS = ((3, 0.3, 4), (-1, -1, -1))
    cfg.TEST.MAX_PER_IMAGE = 50
    cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N = 500, 100
    cache = {}

    def fmap(scale_i, image, chip, net_map):          # (15, 20) map: one FocusChip, two far-apart ones for the odd images
        out = np.zeros_like(np.asarray(net_map, np.float32))
        out[0] = 1.0
        out[1, 1:3, 1:3] = 0.9
        if image % 2:
            out[1, -3:-1, -3:-1] = 0.9
        out[0] -= out[1]
        return out

    def run(lanes, device_compact):
        inference.Tester.device_compact = device_compact
        try:
            roidb = [dict(r) for r in base]
            return inference.imdb_detection_wrapper(rn.resnet_mx_101_e2e, cfg, _Imdb(81), roidb, [mx.gpu(0)], None, None,
                                                    module_cache=cache, focus_map_fn=fmap, return_scale_dets=True, lanes=lanes)
        finally:
            inference.Tester.device_compact = True
    for _ in range(3):
        want_final, want_scales = run(1, False)
        got_final, got_scales = run(3, True)
        assert sum(len(c) for c in got_scales[1][1]) > 6           # some image did get two chips
        for ws, gs in zip(want_scales, got_scales):
            for j in range(1, 81):
                for wi, gi in zip(ws[j], gs[j]):
                    assert len(wi) == len(gi)
                    for wc, gc in zip(wi, gi):
                        assert np.array_equal(np.asarray(wc, np.float64).reshape(-1, 5), np.asarray(gc, np.float64).reshape(-1, 5))
        for j in range(1, 81):
            for wi, gi in zip(want_final[j], got_final[j]):
                assert np.array_equal(wi, gi)
    # the device aggregation's HBM budget: the second scale would exceed it -> its rows come to the host per batch, the first
    # scale's rows are brought over once (_rows_to_host), the host statement aggregates: same final boxes
    import os
    os.environ['SNIPER_DEVICE_AGG_GB'] = '0.003'           # 3.2 MB: the 6 chips of scale 1 (80 classes x 100 RoIs x 40 B each = 1.9 MB) fit, the 9 - 12 of scale 2 do not
    try:
        over_final = inference.imdb_detection_wrapper(rn.resnet_mx_101_e2e, cfg, _Imdb(81), [dict(r) for r in base], [mx.gpu(0)], None, None,
                                                      module_cache=cache, focus_map_fn=fmap, lanes=3)
    finally:
        del os.environ['SNIPER_DEVICE_AGG_GB']
    plain_final = inference.imdb_detection_wrapper(rn.resnet_mx_101_e2e, cfg, _Imdb(81), [dict(r) for r in base], [mx.gpu(0)], None, None,
                                                   module_cache=cache, focus_map_fn=fmap, lanes=3)
    for j in range(1, 81):
        for wi, oi, pi in zip(want_final[j], over_final[j], plain_final[j]):
            assert np.array_equal(wi, oi) and np.array_equal(wi, pi)


def test_device_aggregation_equals_the_host_statement():
    """Tester.aggregate_device (valid-range regrouping, soft-NMS and the MAX_PER_IMAGE rule on the rows sn_det_compact left in HBM)
    == Tester.aggregate's host statement, bit for bit: three scales with both range bounds, chips without rows, classes without
    rows, an image without chips at a scale, tied scores on the MAX_PER_IMAGE boundary, images below and above the cap."""
    from sniper_amd import config as cfgmod
    from sniper_amd import inference
    rs = np.random.RandomState(3)
    NC, n_img = 9, 7                                  # 8 foreground classes
    cfg = cfgmod.res101_e2e_autofocus()
    cfg.TEST.VALID_RANGES = ((30, -1), (-1, 60), (10, 45))
    cfg.TEST.MAX_PER_IMAGE = 40
    dets = []
    for s_i in range(3):
        d = inference._Detections([[[] for _ in range(n_img)] for _ in range(NC)])
        for i in range(n_img):
            n_chips = 0 if (s_i == 2 and i == 4) else 1 + (i + s_i) % 3
            for j in range(NC):
                d[j][i] = [np.zeros((0, 5)) for _ in range(n_chips)]
            for c in range(n_chips):
                lens = rs.randint(0, 9, NC - 1) * (rs.rand(NC - 1) < 0.7)
                if i == 5:
                    lens = np.minimum(lens, 1) * (np.arange(NC - 1) < 3)           # an image that stays below the cap
                if (i + c) % 5 == 0:
                    lens[:] = 0                                                    # a chip without detections
                n = int(lens.sum())
                xy = rs.uniform(0, 300, (n, 2))
                wh = rs.uniform(5, 80, (n, 2))
                sc = np.round(rs.uniform(0.01, 1.0, (n, 1)), 2 if i == 2 else 6)    # image 2: many tied scores
                big = np.ascontiguousarray(np.hstack((xy, xy + wh, sc)), np.float64)
                ends = np.cumsum(lens)
                for j in range(1, NC):
                    d[j][i][c] = big[ends[j - 1] - lens[j - 1]:ends[j - 1]]
                d.compact[(i, c)] = (big, lens.astype(np.int64))
                cap = n + 5                                                        # (the device buffer is a capacity, not a count)
                dev_rows = torch.full((cap, 5), 7.0, dtype=torch.float64, device=dev())
                dev_rows[:n] = torch.from_numpy(big).to(dev())
                d.device_parts[(i, c)] = (dev_rows, torch.from_numpy(lens.astype(np.int32)).to(dev()))
        dets.append(d)

    class Imdb(object):
        num_classes, classes, name, result_path = NC, None, 'synthetic', None
    tester = inference.Tester(None, Imdb(), [{} for _ in range(n_img)], None, cfg=cfg, batch_size=2)
    got = tester.aggregate_device(dets)
    assert got is not None
    for d in dets:
        d.device_parts = {}                                                        # -> the host statement
    want = tester.aggregate(dets)
    capped = 0
    for i in range(n_img):
        n_i = 0
        for j in range(1, NC):
            w, g = np.asarray(want[j][i], np.float32).reshape(-1, 5), np.asarray(got[j][i], np.float32).reshape(-1, 5)
            assert w.shape == g.shape and np.array_equal(w, g), (i, j, w.shape, g.shape)
            n_i += len(g)
        capped += n_i >= cfg.TEST.MAX_PER_IMAGE
        assert n_i > 0
    assert capped >= 3 and sum(len(got[j][5]) for j in range(1, NC)) < cfg.TEST.MAX_PER_IMAGE


def test_rank_sharded_inference_equals_one_process():
    """two ranks (gloo, sharing this one card -- a rehearsal of the control flow) each run every second image through
    both scales, rank 0 gathers and aggregates: final boxes bit-equal to the whole roidb in one process (tests/infer_shard_worker.py;
    the CPU-only twin with stand-in forwards is tests/test_infer_shard_gloo.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONDONTWRITEBYTECODE='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29541', os.path.join(root, 'tests', 'infer_shard_worker.py')]
    r = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and 'SHARD_RESULT ok=1' in r.stdout, r.stdout[-3000:]


def test_inference_forward_graph_replay_equals_eager():
    """A bound test-time executor runs eagerly once, captures its forward on the second call and replays it from then on
    (sniper_amd/engine/executor.py): every output of the replayed graph equals the eager forward of a second Module on the same
    inputs bit for bit, for fresh inputs on every call."""
    import os
    import sniper_amd.mx as mx
    from sniper_amd import config as cfgmod
    from sniper_amd.symbols.faster import resnet_mx_101_e2e as rn
    cfg = cfgmod.res101_e2e_autofocus()
    cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N = 600, 100
    shapes = [('data', (2, 3, 192, 256)), ('im_info', (2, 3)), ('im_ids', (2,)), ('chip_ids', (2,))]

    def module():
        net = rn.resnet_mx_101_e2e(n_proposals=400, test_nbatch=2)
        sym = net.get_symbol_rcnn(cfg, is_train=False)
        net.infer_shape(dict(shapes))
        mod = mx.mod.Module(symbol=sym, context=[mx.gpu(0)], data_names=[k for k, _ in shapes], label_names=None)
        mod.bind(shapes, None, for_training=False)
        return net, mod
    net, graphed = module()
    rs = np.random.RandomState(7)
    arg = {}
    for k, s in net.arg_shape_dict.items():          # MSRA weights, damped residual branches: activations stay inside fp16
        if k in dict(shapes):
            continue
        if k.endswith('_gamma'):
            v = np.full(s, 0.3 if k.endswith('_bn3_gamma') else 1.0, np.float32)
        elif k.endswith('_weight') and len(s) > 1:
            v = (rs.standard_normal(s) * np.sqrt(2.0 / float(np.prod(s[1:])))).astype(np.float32)
        else:
            v = (rs.standard_normal(s) * 0.01).astype(np.float32)
        arg[k] = mx.nd.array(v)
    aux = {k: mx.nd.array(np.full(s, 1600.0 if k == 'bn_data_moving_var' else 1.0, np.float32) if k.endswith('_var')
                          else np.zeros(s, np.float32)) for k, s in net.aux_shape_dict.items()}
    graphed.init_params(arg_params=arg, aux_params=aux)
    old = os.environ.get('SNIPER_HIP_GRAPHS')
    os.environ['SNIPER_HIP_GRAPHS'] = '0'
    try:
        _, eager = module()
        eager.init_params(arg_params=arg, aux_params=aux)
    finally:
        if old is None:
            del os.environ['SNIPER_HIP_GRAPHS']
        else:
            os.environ['SNIPER_HIP_GRAPHS'] = old
    modes = []
    for call in range(4):
        data = [mx.nd.array((rs.standard_normal((2, 3, 192, 256)) * 40).astype(np.float32)),
                mx.nd.array(np.array([[192, 256, 1.0], [180, 256, 0.9]], np.float32)), mx.nd.array(np.array([call, call + 1], np.float32)),
                mx.nd.array(np.zeros(2, np.float32))]
        batch = mx.io.DataBatch(data=data, label=None, pad=0, index=None, provide_data=shapes, provide_label=None)
        graphed.forward(batch, is_train=False)
        a = [o.asnumpy() for o in graphed.get_outputs()]
        eager.forward(batch, is_train=False)
        b = [o.asnumpy() for o in eager.get_outputs()]
        exe = next(iter(graphed._exes.values()))
        modes.append(exe._infer_graph is not None)
        assert len(a) == len(b) >= 6
        for name, x, y in zip(graphed.output_names, a, b):
            assert np.array_equal(x, y), (call, name, float(np.abs(x - y).max()))
        assert np.isfinite(a[1]).all()
    assert modes == [False, True, True, True]
    assert next(iter(eager._exes.values()))._infer_graph is None


def test_inference_batchnorm_folds_into_the_producing_convolution():
    """Test-time executors let a convolution apply the BatchNorm (+ ReLU) that alone reads its output (scaled weights, shift as
    bias, ReLU in the epilogue; sniper_amd/engine/ops.py BatchNormStep.setup / ConvolutionStep.refold).  Same outputs as the
    unfolded graph (SNIPER_INFER_FOLD_BN=0) within fp16 rounding, for 3x3 / 1x1 / strided convolutions with and without bias,
    a BatchNorm without ReLU, and a BatchNorm whose input has a second reader (not folded)."""
    import os
    import sniper_amd.mx as mx
    d = mx.sym.Variable('data')
    c1 = mx.sym.Convolution(data=d, kernel=(3, 3), pad=(1, 1), num_filter=64, no_bias=True, name='c1')
    b1 = mx.sym.BatchNorm(data=c1, fix_gamma=False, eps=2e-5, use_global_stats=True, name='b1')
    r1 = mx.sym.Activation(data=b1, act_type='relu', name='r1')
    c2 = mx.sym.Convolution(data=r1, kernel=(1, 1), num_filter=128, no_bias=False, name='c2')
    b2 = mx.sym.BatchNorm(data=c2, fix_gamma=False, eps=2e-5, use_global_stats=True, name='b2')          # no ReLU behind it
    c3 = mx.sym.Convolution(data=b2, kernel=(3, 3), stride=(2, 2), pad=(1, 1), num_filter=64, no_bias=True, name='c3')
    b3 = mx.sym.BatchNorm(data=c3, fix_gamma=False, eps=2e-5, use_global_stats=True, name='b3')          # c3 has two readers
    r3 = mx.sym.Activation(data=b3, act_type='relu', name='r3')
    out = mx.sym.Group([r3, c3])
    shapes = [('data', (2, 64, 24, 40))]
    rs = np.random.RandomState(5)

    def run(fold):
        old = os.environ.get('SNIPER_INFER_FOLD_BN')
        os.environ['SNIPER_INFER_FOLD_BN'] = '1' if fold else '0'
        try:
            mod = mx.mod.Module(symbol=out, context=[mx.gpu(0)], data_names=['data'], label_names=None)
            mod.bind(shapes, None, for_training=False)
        finally:
            if old is None:
                del os.environ['SNIPER_INFER_FOLD_BN']
            else:
                os.environ['SNIPER_INFER_FOLD_BN'] = old
        return mod
    folded, plain = run(True), run(False)
    exe = next(iter(folded._exes.values()))
    arg, aux = {}, {}
    for name, p in exe.params.items():
        shp = p.ref_shape
        if name.endswith('_gamma'):
            v = rs.uniform(0.5, 1.5, shp)
        elif name.endswith('_weight'):
            v = rs.standard_normal(shp) * np.sqrt(2.0 / np.prod(shp[1:]))
        else:
            v = rs.standard_normal(shp) * 0.3
        arg[name] = mx.nd.array(v.astype(np.float32))
    for name, t in exe.aux.items():
        aux[name] = mx.nd.array((rs.uniform(0.5, 2.0, tuple(t.shape)) if name.endswith('_var') else rs.standard_normal(tuple(t.shape)) * 0.2
                                 ).astype(np.float32))
    for m in (folded, plain):
        m.init_params(arg_params=arg, aux_params=aux)
    kinds = {type(s).__name__ + ':' + s.node.name: getattr(s, 'folded_into', None) is not None for s in exe.steps
             if type(s).__name__ == 'BatchNormStep'}
    assert kinds == {'BatchNormStep:b1': True, 'BatchNormStep:b2': True, 'BatchNormStep:b3': False}, kinds
    assert all(getattr(s, 'folded_into', None) is None for s in next(iter(plain._exes.values())).steps)
    for call in range(3):            # eager, capture, replay
        x = mx.nd.array(rs.standard_normal(shapes[0][1]).astype(np.float32))
        batch = mx.io.DataBatch(data=[x], label=None, pad=0, index=None, provide_data=shapes, provide_label=None)
        folded.forward(batch, is_train=False)
        a = [o.asnumpy() for o in folded.get_outputs()]
        plain.forward(batch, is_train=False)
        b = [o.asnumpy() for o in plain.get_outputs()]
        for u, v in zip(a, b):
            assert u.shape == v.shape and np.isfinite(u).all()
            err = np.abs(u - v).max() / max(np.abs(v).max(), 1e-6)
            assert err < 4e-3, (call, err)         # two fp16 roundings of a scaled weight vs of the normalised output
        assert (a[0] >= 0).all() and (a[0] > 0).mean() > 0.2
    # new parameters after the capture reach the folded buffers (set_params -> params_changed -> refold)
    arg2 = {k: mx.nd.array(v.asnumpy() * 0.5) if k.endswith('_gamma') else v for k, v in arg.items()}
    for m in (folded, plain):
        m.set_params(arg2, aux)
    folded.forward(batch, is_train=False)
    plain.forward(batch, is_train=False)
    u, v = folded.get_outputs()[0].asnumpy(), plain.get_outputs()[0].asnumpy()
    assert np.abs(u - v).max() / max(np.abs(v).max(), 1e-6) < 4e-3 and not np.allclose(u, a[0])


def test_inference_unit_opening_batchnorm_is_the_second_output_of_the_residual_convolution():
    """Test-time pre-activation units: the BatchNorm + ReLU that opens unit k + 1 reads the residual sum unit k's last convolution
    writes (two readers: it cannot fold) and is that convolution's SECOND output (sn_conv_fwd_dual).  Two chained units with an
    identity and a projection shortcut, a large batch (plain launch) and a 2-chip batch (split-K launch + dual reduce): bit-equal
    to the separate sn_bn_apply (SNIPER_INFER_DUAL_BN=0) on the eager, capturing and replaying calls, and after set_params."""
    import os
    import sniper_amd.mx as mx

    def bn_relu(x, name):
        return mx.sym.Activation(data=mx.sym.BatchNorm(data=x, fix_gamma=False, eps=2e-5, use_global_stats=True, name=name + '_bn'),
                                 act_type='relu', name=name + '_relu')

    def unit(x, nf, name, match):
        a1 = bn_relu(x, name + '1')
        c1 = mx.sym.Convolution(data=a1, kernel=(1, 1), num_filter=nf // 4, no_bias=True, name=name + '_c1')
        c2 = mx.sym.Convolution(data=bn_relu(c1, name + '2'), kernel=(3, 3), pad=(1, 1), num_filter=nf // 4, no_bias=True, name=name + '_c2')
        c3 = mx.sym.Convolution(data=bn_relu(c2, name + '3'), kernel=(1, 1), num_filter=nf, no_bias=True, name=name + '_c3')
        sc = x if match else mx.sym.Convolution(data=a1, kernel=(1, 1), num_filter=nf, no_bias=True, name=name + '_sc')
        return c3 + sc
    d = mx.sym.Variable('data')
    u1 = unit(d, 512, 'u1', False)
    u2 = unit(u1, 512, 'u2', True)
    u3 = unit(u2, 512, 'u3', True)
    out = bn_relu(u3, 'top')          # (a residual sum that is ALSO a graph output is fp32: its BatchNorm keeps the separate launch)
    rs = np.random.RandomState(6)
    for shape in ((8, 256, 32, 40), (2, 256, 12, 20)):
        shapes = [('data', shape)]

        def run(dual):
            old = os.environ.get('SNIPER_INFER_DUAL_BN')
            os.environ['SNIPER_INFER_DUAL_BN'] = '1' if dual else '0'
            try:
                mod = mx.mod.Module(symbol=out, context=[mx.gpu(0)], data_names=['data'], label_names=None)
                mod.bind(shapes, None, for_training=False)
            finally:
                if old is None:
                    del os.environ['SNIPER_INFER_DUAL_BN']
                else:
                    os.environ['SNIPER_INFER_DUAL_BN'] = old
            return mod
        fused, plain = run(True), run(False)
        exe = next(iter(fused._exes.values()))
        arg, aux = {}, {}
        for name, p in exe.params.items():
            shp = p.ref_shape
            v = rs.uniform(0.5, 1.5, shp) if name.endswith('_gamma') else \
                (rs.standard_normal(shp) * np.sqrt(2.0 / np.prod(shp[1:])) if name.endswith('_weight') else rs.standard_normal(shp) * 0.3)
            arg[name] = mx.nd.array(v.astype(np.float32))
        for name, t in exe.aux.items():
            aux[name] = mx.nd.array((rs.uniform(0.5, 2.0, tuple(t.shape)) if name.endswith('_var') else
                                     rs.standard_normal(tuple(t.shape)) * 0.2).astype(np.float32))
        for m in (fused, plain):
            m.init_params(arg_params=arg, aux_params=aux)
        dual = sorted(s.node.name for s in exe.steps if type(s).__name__ == 'BatchNormStep' and getattr(s, 'dual_from', None) is not None)
        # the BatchNorm that opens u2, u3 and the one on top read a residual sum; u1's reads the graph input (no producing convolution)
        assert dual == ['top_bn', 'u21_bn', 'u31_bn'], dual
        assert not any(getattr(s, 'dual_from', None) is not None for s in next(iter(plain._exes.values())).steps
                       if type(s).__name__ == 'BatchNormStep')
        for call in range(3):            # eager, capture, replay
            x = mx.nd.array(rs.standard_normal(shape).astype(np.float32))
            batch = mx.io.DataBatch(data=[x], label=None, pad=0, index=None, provide_data=shapes, provide_label=None)
            fused.forward(batch, is_train=False)
            a = [o.asnumpy() for o in fused.get_outputs()]
            plain.forward(batch, is_train=False)
            b = [o.asnumpy() for o in plain.get_outputs()]
            for u, v in zip(a, b):
                assert np.isfinite(u).all() and np.array_equal(u, v), (shape, call, float(np.abs(u - v).max()))
            assert (a[0] > 0).mean() > 0.2
        arg2 = {k: mx.nd.array(v.asnumpy() * 0.5) if k.endswith('_gamma') else v for k, v in arg.items()}
        for m in (fused, plain):
            m.set_params(arg2, aux)
        fused.forward(batch, is_train=False)
        plain.forward(batch, is_train=False)
        u, v = fused.get_outputs()[0].asnumpy(), plain.get_outputs()[0].asnumpy()
        assert np.array_equal(u, v) and not np.allclose(u, a[0])


def test_set_params_after_capture_reaches_unfolded_batch_statistics_layers():
    """a test-time executor normalises use_global_stats=False layers with the moving statistics too; when such a
    layer is NOT folded into its producer (here: its input has a second reader), its scale / shift used to be recomputed
    lazily in forward() -- which a hipGraph replay never reaches.  New parameters after the capture must change the replayed
    output exactly as they change an eager executor's."""
    import os
    import sniper_amd.mx as mx
    d = mx.sym.Variable('data')
    c1 = mx.sym.Convolution(data=d, kernel=(1, 1), num_filter=64, no_bias=True, name='c1')
    b1 = mx.sym.BatchNorm(data=c1, fix_gamma=False, eps=2e-5, use_global_stats=False, name='b1')     # c1 has two readers: not folded
    r1 = mx.sym.Activation(data=b1, act_type='relu', name='r1')
    out = mx.sym.Group([r1, c1])
    shapes = [('data', (2, 64, 16, 24))]
    rs = np.random.RandomState(11)

    def module(graphs):
        old = os.environ.get('SNIPER_HIP_GRAPHS')
        os.environ['SNIPER_HIP_GRAPHS'] = '1' if graphs else '0'
        try:
            mod = mx.mod.Module(symbol=out, context=[mx.gpu(0)], data_names=['data'], label_names=None)
            mod.bind(shapes, None, for_training=False)
        finally:
            if old is None:
                del os.environ['SNIPER_HIP_GRAPHS']
            else:
                os.environ['SNIPER_HIP_GRAPHS'] = old
        return mod
    graphed, eager = module(True), module(False)
    exe = next(iter(graphed._exes.values()))
    assert all(getattr(s, 'folded_into', None) is None for s in exe.steps)
    arg = {'c1_weight': mx.nd.array((rs.standard_normal((64, 64, 1, 1)) * 0.2).astype(np.float32)),
           'b1_gamma': mx.nd.array(rs.uniform(0.5, 1.5, (64,)).astype(np.float32)),
           'b1_beta': mx.nd.array((rs.standard_normal((64,)) * 0.3).astype(np.float32))}
    aux = {'b1_moving_mean': mx.nd.array((rs.standard_normal((64,)) * 0.2).astype(np.float32)),
           'b1_moving_var': mx.nd.array(rs.uniform(0.5, 2.0, (64,)).astype(np.float32))}
    for m in (graphed, eager):
        m.init_params(arg_params=arg, aux_params=aux)
    x = mx.nd.array(rs.standard_normal(shapes[0][1]).astype(np.float32))
    batch = mx.io.DataBatch(data=[x], label=None, pad=0, index=None, provide_data=shapes, provide_label=None)
    for call in range(3):            # eager, capture, replay
        graphed.forward(batch, is_train=False)
        eager.forward(batch, is_train=False)
    before = graphed.get_outputs()[0].asnumpy().copy()
    assert exe._infer_graph is not None
    assert np.array_equal(before, eager.get_outputs()[0].asnumpy())
    arg2 = dict(arg, b1_gamma=mx.nd.array(arg['b1_gamma'].asnumpy() * 0.25), b1_beta=mx.nd.array(arg['b1_beta'].asnumpy() + 0.5))
    aux2 = dict(aux, b1_moving_mean=mx.nd.array(aux['b1_moving_mean'].asnumpy() - 0.3))
    for m in (graphed, eager):
        m.set_params(arg2, aux2)
    graphed.forward(batch, is_train=False)
    eager.forward(batch, is_train=False)
    after, want = graphed.get_outputs()[0].asnumpy(), eager.get_outputs()[0].asnumpy()
    assert exe._infer_graph is not None          # still the replayed graph
    assert np.array_equal(after, want)
    assert not np.allclose(after, before)


def _bound_test_module(bind, cfg, seed):
    """R101 test graph bound at `bind`, MSRA weights with damped residual branches (activations stay inside fp16)."""
    import sniper_amd.mx as mx
    from sniper_amd.symbols.faster import resnet_mx_101_e2e as rn

    def module():
        net = rn.resnet_mx_101_e2e(n_proposals=400, test_nbatch=bind[0][1][0])
        sym = net.get_symbol_rcnn(cfg, is_train=False)
        net.infer_shape(dict(bind))
        mod = mx.mod.Module(symbol=sym, context=[mx.gpu(0)], data_names=[k for k, _ in bind], label_names=None)
        mod.bind(bind, None, for_training=False)
        return net, mod
    net, first = module()
    rs = np.random.RandomState(seed)
    arg = {}
    for k, s in net.arg_shape_dict.items():
        if k in dict(bind):
            continue
        if k.endswith('_gamma'):
            v = np.full(s, 0.3 if k.endswith('_bn3_gamma') else 1.0, np.float32)
        elif k.endswith('_weight') and len(s) > 1:
            v = (rs.standard_normal(s) * np.sqrt(2.0 / float(np.prod(s[1:])))).astype(np.float32)
        else:
            v = (rs.standard_normal(s) * 0.01).astype(np.float32)
        arg[k] = mx.nd.array(v)
    aux = {k: mx.nd.array(np.full(s, 1600.0 if k == 'bn_data_moving_var' else 1.0, np.float32) if k.endswith('_var')
                          else np.zeros(s, np.float32)) for k, s in net.aux_shape_dict.items()}
    first.init_params(arg_params=arg, aux_params=aux)

    def another():
        _, mod = module()
        mod.init_params(arg_params=arg, aux_params=aux)
        return mod
    return first, another, rs


def _test_batch(bind, h, w, rs, tag):
    import sniper_amd.mx as mx
    shp = [('data', (2, 3, h, w))] + bind[1:]
    data = [mx.nd.array((rs.standard_normal((2, 3, h, w)) * 40).astype(np.float32)),
            mx.nd.array(np.array([[h, w, 1.0], [h - 12, w, 0.9]], np.float32)),
            mx.nd.array(np.array([tag, tag + 1], np.float32)), mx.nd.array(np.zeros(2, np.float32))]
    return mx.io.DataBatch(data=data, label=None, pad=0, index=None, provide_data=shp, provide_label=None)


def test_executor_cache_eviction_rebuilds_and_stays_correct(monkeypatch):
    """SNIPER_EXE_CACHE=2 and three batch shapes in turn: every visit is a miss (the least recently used shape is dropped, its
    cycles collected after gc.unfreeze, the shape rebuilt into the shared pool on its next visit) -- outputs equal those of a
    Module that keeps all three, bit for bit."""
    from sniper_amd import config as cfgmod
    cfg = cfgmod.res101_e2e_autofocus()
    cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N = 600, 100
    bind = [('data', (2, 3, 256, 320)), ('im_info', (2, 3)), ('im_ids', (2,)), ('chip_ids', (2,))]
    keeps, another, rs = _bound_test_module(bind, cfg, 13)
    monkeypatch.setenv('SNIPER_EXE_CACHE', '2')
    small = another()
    built = []
    for rnd in range(3):
        for h, w in [(256, 320), (128, 192), (192, 256)]:
            batch = _test_batch(bind, h, w, rs, rnd)
            small.forward(batch, is_train=False)
            a = [o.asnumpy() for o in small.get_outputs()]
            built.append(id(small.exe))
            assert len(small._exes) <= 2
            monkeypatch.delenv('SNIPER_EXE_CACHE')
            keeps.forward(batch, is_train=False)
            monkeypatch.setenv('SNIPER_EXE_CACHE', '2')
            b = [o.asnumpy() for o in keeps.get_outputs()]
            for name, x, y in zip(small.output_names, a, b):
                assert np.array_equal(x, y), (rnd, (h, w), name)
    assert len(keeps._exes) == 3
    assert len(small._act_pool.buffers) == len(keeps._act_pool.buffers)      # rebuilt shapes went back into the same pool


def test_new_shape_of_a_warm_module_captures_on_its_first_forward(monkeypatch):
    """A batch shape a warm test-time Module has not met: its executor adopts every derived buffer of the Module, so its FIRST forward
    is the hipGraph capture (Module._exe_for: capture_first) -- outputs of that first call and of the replay behind it equal, bit for
    bit, those of a Module that runs new shapes eagerly first (SNIPER_CAPTURE_FIRST=0)."""
    from sniper_amd import config as cfgmod
    cfg = cfgmod.res101_e2e_autofocus()
    cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N = 600, 100
    bind = [('data', (2, 3, 256, 320)), ('im_info', (2, 3)), ('im_ids', (2,)), ('chip_ids', (2,))]
    fast, another, rs = _bound_test_module(bind, cfg, 17)
    slow = another()
    warm = _test_batch(bind, 256, 320, rs, 0)
    for m in (fast, slow):
        for _ in range(3):
            m.forward(warm, is_train=False)
    for k, (h, w) in enumerate([(128, 192), (192, 256)]):
        for rep in range(3):
            batch = _test_batch(bind, h, w, rs, 10 * k + rep)
            monkeypatch.delenv('SNIPER_CAPTURE_FIRST', raising=False)
            fast.forward(batch, is_train=False)
            a = [o.asnumpy() for o in fast.get_outputs()]
            if rep == 0:
                assert fast.exe._infer_graph is not None and fast.exe._infer_calls == 1      # captured by the first call
            monkeypatch.setenv('SNIPER_CAPTURE_FIRST', '0')
            slow.forward(batch, is_train=False)
            b = [o.asnumpy() for o in slow.get_outputs()]
            if rep == 0:
                assert slow.exe._infer_graph is None                                         # ran eagerly
            for name, x, y in zip(fast.output_names, a, b):
                assert np.array_equal(x, y), ((h, w), rep, name)


def test_bound_shapes_of_one_module_share_their_activation_memory():
    """Test-time executors of one Module lie over the same activation bytes (engine/executor.py::ActivationPool) and hold ONE set
    of parameters (share_params).  Three batch shapes visited in turn, five rounds (eager, capture, replays interleaved), then
    new parameters loaded into the Module and two more rounds: every output equals, bit for bit, the output of a Module whose
    executors own their activations and parameters (SNIPER_SHARE_ACTIVATIONS=0, SNIPER_SHARE_PARAMS=0) -- no executor depends on
    bytes another one overwrote, every shape sees a parameter load -- and the pool holds the largest shape's footprint, not the sum."""
    import os
    import sniper_amd.mx as mx
    from sniper_amd import config as cfgmod
    cfg = cfgmod.res101_e2e_autofocus()
    cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N = 600, 100
    bind = [('data', (2, 3, 256, 320)), ('im_info', (2, 3)), ('im_ids', (2,)), ('chip_ids', (2,))]
    sizes = [(256, 320), (128, 192), (192, 256)]
    shared, another, rs = _bound_test_module(bind, cfg, 11)
    old = {k: os.environ.get(k) for k in ('SNIPER_SHARE_ACTIVATIONS', 'SNIPER_SHARE_PARAMS')}
    os.environ['SNIPER_SHARE_ACTIVATIONS'] = os.environ['SNIPER_SHARE_PARAMS'] = '0'
    try:
        own = another()
        for h, w in sizes:                 # bind the unshared Module's executors while the switches are off
            full = dict([('data', (2, 3, h, w))] + bind[1:])
            own._exe_for({k: full[k] for k in own.exe.input_names})
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v

    def rounds(n, tag0):
        for rnd in range(n):
            for h, w in sizes:
                batch = _test_batch(bind, h, w, rs, tag0 + rnd)
                shared.forward(batch, is_train=False)
                a = [o.asnumpy() for o in shared.get_outputs()]
                own.forward(batch, is_train=False)
                b = [o.asnumpy() for o in own.get_outputs()]
                for name, x, y in zip(shared.output_names, a, b):
                    assert np.array_equal(x, y), (tag0 + rnd, (h, w), name)
                assert np.isfinite(a[1]).all()
    rounds(5, 0)
    assert all(e.act_pool is None and not e.shared_names for e in own._exes.values())
    assert all(e.act_pool is shared._act_pool for e in shared._exes.values())
    assert len(shared._exes) == len(own._exes) == 3
    assert all(e._infer_graph is not None for e in shared._exes.values())
    exes = list(shared._exes.values())
    name = 'stage3_unit1_conv2_weight'
    assert len({e.params[name].w16.data_ptr() for e in exes}) == 1 and len({e.params[name].master.data_ptr() for e in exes}) == 1
    assert len({e.aux['stage3_unit1_bn2_moving_var'].data_ptr() for e in exes}) == 1
    assert len({e.params[name].w16.data_ptr() for e in own._exes.values()}) == 3
    # a parameter load after the captures reaches every bound shape (shared masters, per-shape derived buffers)
    arg, aux = shared.get_params()
    r2 = np.random.RandomState(5)
    arg2 = {k: mx.nd.array(v.asnumpy() * (1.0 + 0.05 * r2.standard_normal(v.shape).astype(np.float32))) for k, v in arg.items()}
    aux2 = {k: mx.nd.array(v.asnumpy() * (1.3 if k.endswith('_var') else 1.0) + (0.0 if k.endswith('_var') else 0.01))
            for k, v in aux.items()}
    shared.set_params(arg2, aux2)
    own.set_params(arg2, aux2)
    before = [o.asnumpy().copy() for o in shared.get_outputs()]
    rounds(2, 100)
    assert getattr(own, '_act_pool', None) is None
    pool = shared._act_pool
    lo = pool.buffers[0].data_ptr() + (-pool.buffers[0].data_ptr()) % pool.ALIGN
    for e in shared._exes.values():
        ptrs = [v.t.data_ptr() for v in e.vals.values() if v.t is not None and v.producer is not None]
        inside = [q for q in ptrs if any(b.data_ptr() <= q < b.data_ptr() + b.numel() for b in pool.buffers)]
        assert lo in ptrs and len(inside) > 100             # every shape starts at the pool's first byte
    del before



def test_dropped_modules_release_their_memory():
    """Bound executors leave the cyclic collector's reach (engine/executor.py::settle_heap); a Module that is dropped -- the
    reference rebuilds one per scale and call (lib/inference.py:411-436) -- must still give its pool, parameters and executors
    back: the next bind thaws and collects first (thaw_heap)."""
    import gc
    from sniper_amd import config as cfgmod
    cfg = cfgmod.res101_e2e_autofocus()
    cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N = 600, 100
    bind = [('data', (2, 3, 256, 320)), ('im_info', (2, 3)), ('im_ids', (2,)), ('chip_ids', (2,))]
    rs = np.random.RandomState(3)
    held = []
    for rnd in range(4):
        mod, _, _ = _bound_test_module(bind, cfg, 17)
        for _ in range(3):                                   # eager, capture, replay
            mod.forward(_test_batch(bind, 256, 320, rs, rnd), is_train=False)
            [o.asnumpy() for o in mod.get_outputs()]
        assert gc.get_freeze_count() > 0                     # the executor's objects are out of the collector's reach
        del mod
        torch.cuda.synchronize()
        held.append(torch.cuda.memory_allocated())
    # every round starts by collecting the previous round's Module: what is held does not grow with the rounds
    assert held[3] <= held[1] + (64 << 20), held
