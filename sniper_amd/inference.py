"""Inference driver, contract of lib/inference.py (`Tester` :26-408, `detect_scale_worker` :411-436,
`imdb_detection_wrapper` :439-529): forward -> decode/clip/rescale -> score threshold -> per-class (soft-)NMS ->
AutoFocus pruning / FocusChips -> multi-scale aggregation.

Same class, method names, arguments and return structures (`all_boxes[class][image][chip]` -> `(n,5)` arrays).  What
moved to the GPU (MI355X first, not a translation of the host loops):
  * `detect`: bbox_pred + clip_boxes + /scale of every chip of the batch is ONE `sn_bbox_decode` launch
    (reference: a numpy loop per chip, :122-131);
  * `nms_worker` / `aggregate`: the 80 classes x images of independent soft-NMS problems are batched into
    `sn_soft_nms_batch` launches (reference: `Pool(32).map` over `cpu_soft_nms`, :159-201; ThreadPool(8) :308-310);
  * hard NMS uses the bitmask kernel.
Dataset I/O (imdb, pickled caches, visualisation) is out of scope: `imdb` only needs `result_path`, `num_classes`,
`classes`, `name`."""
import math
import os
import threading

import numpy as np
import torch

import sniper_amd.mx as mx

from . import hip
from .chips_inference import add_chips
from .ext import cpu_nms as _cpu_nms
from .ext import gpu_nms as _gpu_nms
from .iterators.MNIteratorTestAutoFocus import MNIteratorTestAutoFocus
from .iterators.PrefetchingIter import PrefetchingIter


_PINNED = {}     # (job slot, (lane, ping / pong, gpu, output), shape, dtype) -> pinned host tensor of Tester._launch
_LANE_STREAMS = {}        # job slot -> the side streams of Tester's lanes (created once: streams are not free to make)
_SLOT = threading.local()       # .job: which concurrent job of imdb_detection_wrapper this thread runs (0 outside of it).  Buffers and
                                # streams are keyed by the SLOT, not the thread: the job threads of every call are new ones



# RoIs per chip of the test-time graphs (detect_scale_worker builds the symbol with it; the reference: lib/inference.py:416,537)
TEST_N_PROPOSALS = 400

class nms_wrapper(object):
    """lib/nms/nms.py:15-23: hard NMS when thresh > 0, gaussian soft-NMS (sigma) otherwise."""

    def __init__(self, thresh, sigma):
        assert thresh < 0 or sigma < 0, 'Either nms sigma or nms thresh should be set to negative'
        self.thresh, self.sigma = thresh, sigma

    def process(self, dets):
        return self.process_many([dets])[0]

    def process_many(self, problems):
        """One launch for a list of independent (n,5) problems."""
        problems = [np.ascontiguousarray(d, np.float32).reshape(-1, 5) for d in problems]
        if self.thresh > 0:
            out = []
            for d in problems:
                keep = _gpu_nms.gpu_nms(d, self.thresh) if d.shape[0] else []
                out.append(d[keep, :] if len(keep) else np.zeros((0, 5), np.float32))
            return out
        return _cpu_nms.soft_nms_batch(problems, sigma=self.sigma, Nt=0.3, threshold=0.001, method=2)


    def process_stacked(self, rows, sizes):
        """process_many for problems given as one (total, 5) float32 array + rows per problem."""
        if self.thresh <= 0:
            return _cpu_nms.soft_nms_stacked(rows, sizes, sigma=self.sigma, Nt=0.3, threshold=0.001, method=2)
        return self.process_many(_split_rows(rows, sizes))


class nms_worker(object):
    """lib/data_utils/data_workers.py:124-129."""

    def __init__(self, nms_thresh, nms_sigma):
        self.nms_wrapper = nms_wrapper(nms_thresh, nms_sigma)

    def worker(self, data):
        return self.nms_wrapper.process(data)

    def worker_many(self, datas):
        return self.nms_wrapper.process_many(datas)


def _valid_range_filter(cls_dets, valid_range):
    """lib/inference.py:176-186 (the names are swapped there too: `heights` is the x extent)."""
    heights = cls_dets[:, 2] - cls_dets[:, 0]
    widths = cls_dets[:, 3] - cls_dets[:, 1]
    areas = widths * heights
    ok = np.ones(len(areas), bool)
    if valid_range[0] > 0:
        ok &= areas > valid_range[0] * valid_range[0]
    if valid_range[1] > 0:
        ok &= areas <= valid_range[1] * valid_range[1]
    return cls_dets[ok, :]


def _stack_rows(per_class, dtype):
    """(rows of every class stacked, rows per class) for a list of (n, 5) arrays / empty lists, as `dtype`."""
    lens = np.fromiter((len(a) for a in per_class), np.int64, len(per_class))
    if lens.sum() == 0:
        return np.zeros((0, 5), dtype), lens
    try:
        big = np.concatenate(per_class)            # one pass when every entry already is an (n, 5) array
        if big.ndim != 2 or big.shape[1] != 5:
            raise ValueError
        big = big.astype(dtype, copy=False)          # (concatenate already returned a fresh array: callers may write into it)
    except ValueError:                               # ragged input (the reference leaves [] for "no detections")
        big = np.concatenate([np.asarray(a, dtype).reshape(-1, 5) for a in per_class])
    return big, lens


def _split_rows(rows, counts):
    """np.split(rows, cumsum(counts)[:-1]) as plain slices (np.split spends 2 us per piece on axis bookkeeping; a chip has 80)."""
    ends = np.cumsum(counts).tolist()
    return [rows[a:b] for a, b in zip([0] + ends[:-1], ends)]


def threshold_detections(cscores, cboxes, cls_thresh, num_classes):
    """Tester.get_detections' score threshold for every class of one chip (lib/inference.py:289-295): list over classes
    1..num_classes-1 of `hstack(cboxes[inds, 0:4], cscores[inds, j, None])`, inds = where(cscores[:, j] > cls_thresh) -- one mask,
    one gather and one split instead of a where / hstack pair per class."""
    mask = np.ascontiguousarray((cscores[:, 1:num_classes] > cls_thresh).T)   # (classes, RoIs): a class's RoIs in order
    cls_idx, roi_idx = np.divmod(np.flatnonzero(mask), mask.shape[1])        # (flatnonzero: 5x faster than the 2-D nonzero)
    dets = np.empty((len(roi_idx), 5), np.result_type(cboxes.dtype, cscores.dtype))
    dets[:, 0:4] = cboxes[roi_idx, 0:4]
    dets[:, 4] = cscores[roi_idx, cls_idx + 1]
    return _split_rows(dets, mask.sum(1))


def prune_chip_border(per_class, crop, im_width, im_height):
    """AutoFocus pruning of one chip's detections (lib/inference.py:336-353): translate every class's rows into image coordinates
    and drop those within 10 px of a chip border that is not an image border (`check_valid`, :236-259).  All classes in one
    float64 array, one mask; returns the list over classes of (n, 5) float64 arrays."""
    nc = len(per_class)
    big, lens = _stack_rows(per_class, np.float64)          # a copy: the inputs keep their chip coordinates
    big[:, 0] += crop[0]; big[:, 2] += crop[0]
    big[:, 1] += crop[1]; big[:, 3] += crop[1]
    ok = Tester._valid_mask(big, crop, im_width, im_height)
    cls_id = np.repeat(np.arange(nc), lens)
    kept = np.bincount(cls_id[ok], minlength=nc)
    return _split_rows(big[ok], kept)


def cap_detections_per_image(per_class, max_per_image):
    """The MAX_PER_IMAGE rule of Tester.aggregate (lib/inference.py:203-211) for one image: with more than `max_per_image`
    detections over all classes, every class keeps its rows whose score reaches the max_per_image-th best score of the image
    (ties stay).  -> the per-class arrays after the rule, or None when nothing has to go.  One stacked mask per image instead of
    a `where` + gather per class (10 240 numpy calls per 64-image pass in the per-class form)."""
    lens = [len(d) for d in per_class]
    total = sum(lens)
    if total <= max_per_image:
        return None
    rows = np.concatenate([d for d in per_class if len(d)])
    scores = rows[:, -1]
    thresh = np.partition(scores, total - max_per_image)[total - max_per_image]      # = np.sort(scores)[-max_per_image]
    ok = scores >= thresh
    cls = np.repeat(np.arange(len(per_class)), lens)
    ends = np.cumsum(np.bincount(cls[ok], minlength=len(per_class))).tolist()
    rows = rows[ok]
    out, a = [], 0
    for b in ends:
        out.append(rows[a:b])
        a = b
    return out


def aggregate_problems(scale_cls_dets, valid_ranges, num_images, num_classes, stacked=False):
    """The per (image, class) NMS problems of Tester.aggregate (lib/inference.py:170-190): for image i and class j the rows of
    every scale's every chip that pass that scale's valid range, in (scale, chip, row) order.  Built per image with a handful
    of array operations instead of classes x scales x chips Python iterations: every (scale, chip) part is stacked class-major
    and masked once, and since each part already is class-major the regrouping by class is a block permutation computed from the
    per (part, class) counts -- no sort.  Returns the problems in (image, class) order, float32 (n, 5); stacked=True: the same
    rows as ONE (total, 5) array + the rows per problem (what the batched soft-NMS launch uploads)."""
    nc = num_classes - 1
    if stacked:
        native = _aggregate_stacked_native(scale_cls_dets, valid_ranges, num_images, nc)
        if native is not None:
            return native
    problems, sizes = [], []
    for i in range(num_images):
        parts, counts = [], []
        for all_cls_dets, vr in zip(scale_cls_dets, valid_ranges):
            ready = getattr(all_cls_dets, 'compact', {})
            for c in range(len(all_cls_dets[1][i])):
                if (i, c) in ready:              # rows of this chip already stacked class-major (Tester.get_detections)
                    big, lens = ready[(i, c)]
                    big = big.astype(np.float32)
                else:
                    big, lens = _stack_rows([all_cls_dets[j][i][c] for j in range(1, num_classes)], np.float32)
                if len(big) == 0:
                    continue
                areas = (big[:, 3] - big[:, 1]) * (big[:, 2] - big[:, 0])      # float32 products, as _valid_range_filter
                ok = np.ones(len(big), bool)
                if vr[0] > 0:
                    ok &= areas > vr[0] * vr[0]
                if vr[1] > 0:
                    ok &= areas <= vr[1] * vr[1]
                parts.append(big[ok])
                counts.append(np.bincount(np.repeat(np.arange(nc), lens)[ok], minlength=nc))
        if not parts:
            if stacked:
                sizes.append(np.zeros(nc, np.int64))
            else:
                problems.extend(np.empty((0, 5), np.float32) for _ in range(nc))
            continue
        big = np.concatenate(parts)                  # rows ordered (part, class, row)
        cnt = np.stack(counts)                       # (parts, classes)
        src = np.cumsum(cnt.ravel()) - cnt.ravel()   # first row of block (part, class) in `big`
        length = cnt.T.ravel()                       # blocks in destination order: class-major, parts (scale, chip) inside
        start = src.reshape(cnt.shape).T.ravel()
        total = int(length.sum())
        order = np.repeat(start - (np.cumsum(length) - length), length) + np.arange(total)
        if stacked:
            problems.append(big[order])
            sizes.append(cnt.sum(0))
        else:
            problems.extend(_split_rows(big[order], cnt.sum(0)))
    if stacked:
        rows = np.concatenate(problems) if problems else np.zeros((0, 5), np.float32)
        return rows, (np.concatenate(sizes) if sizes else np.zeros(0, np.int64))
    return problems


def _aggregate_stacked_native(scale_cls_dets, valid_ranges, num_images, nc):
    """aggregate_problems(stacked=True) in one pass of sn_aggregate_problems_host, when EVERY chip's rows are at hand in the form
    the GPU returned them (`compact`: rows grouped by class + rows per class).  None otherwise (the numpy statement runs)."""
    parts, lens, ranges, part_of_image = [], [], [], [0]
    for i in range(num_images):
        for all_cls_dets, vr in zip(scale_cls_dets, valid_ranges):
            ready = getattr(all_cls_dets, 'compact', None)
            n_chips = len(all_cls_dets[1][i]) if len(all_cls_dets) > 1 else 0
            for c in range(n_chips):
                hit = ready.get((i, c)) if ready else None
                if hit is None or hit[0].dtype != np.float64 or not hit[0].flags['C_CONTIGUOUS'] or len(hit[1]) != nc:
                    return None
                parts.append(hit[0])
                lens.append(hit[1])
                # `areas > vr * vr` against a float32 array compares in float32 (numpy's weak Python scalars)
                ranges.append((np.float32(vr[0] * vr[0]) if vr[0] > 0 else np.float32(0), np.float32(vr[1] * vr[1]) if vr[1] > 0 else np.float32(0)))
        part_of_image.append(len(parts))
    P = len(parts)
    sizes = np.zeros(num_images * nc, np.int64)
    if P == 0:
        return np.zeros((0, 5), np.float32), sizes
    lens_a = np.ascontiguousarray(np.stack(lens), np.int64)
    ptrs = np.fromiter((a.ctypes.data for a in parts), np.uint64, P)
    cap = int(lens_a.sum())
    rows, total = np.empty((cap, 5), np.float32), np.zeros(1, np.int64)
    hip.call('sn_aggregate_problems_host', ptrs, lens_a, np.asarray(part_of_image, np.int32), np.asarray(ranges, np.float32).reshape(-1, 2),
             P, nc, num_images, rows, cap, sizes, total)
    return rows[:int(total[0])], sizes


class _Detections(list):
    """all_boxes[class][image][chip] of Tester.get_detections, plus `compact`: {(image, chip): (rows grouped by class, rows per
    class)} for the chips whose rows came back from the GPU already in that form (what aggregate_problems stacks anyway)."""

    def __init__(self, *a):
        super(_Detections, self).__init__(*a)
        self.compact = {}
        # {(image, chip): (device rows (cap, 5) float64 grouped by class, device rows-per-class (nc,) int32)}: what sn_det_compact
        # left in HBM for that chip -- the input of the device aggregation (Tester.aggregate_device)
        self.device_parts = {}

    def __reduce__(self):              # (pickled for the per-scale caches / gathered across ranks: host content only)
        return (_rebuild_detections, (list(self), self.compact))


def _rebuild_detections(items, compact):
    d = _Detections(items)
    d.compact = compact
    return d


class Tester(object):
    device_compact = True        # threshold + prune on the GPU (sn_det_compact); False: the numpy statement of the same (tests)
    # rows of a compacting launch: keep them in HBM for the device aggregation (keep_device_rows) and / or bring them to the host
    # (host_rows: the per-scale detection lists, pickles, visualisation, the host aggregation).  imdb_detection_wrapper sets both.
    keep_device_rows = False
    host_rows = True

    def __init__(self, module, imdb, roidb, test_iter, cfg, rcnn_output_names=None, rpn_output_names=None, logger=None,
                 batch_size=None):
        self.test_iter = test_iter
        self._own_iter = False
        if test_iter is not None and not isinstance(test_iter, PrefetchingIter):
            self.test_iter = PrefetchingIter(self.test_iter, depth=max(2, len(module)) if isinstance(module, (list, tuple)) else 2,
                                             own_stream=False)
            self._own_iter = True
            self.scale = test_iter.test_scale
        self.cfg = cfg
        # `module` may be a list of identical bound Modules ("lanes"): batch k runs on lane k % len(lanes), each lane on its own
        # HIP stream, so up to that many forwards are in flight (get_detections)
        self.modules = list(module) if isinstance(module, (list, tuple)) else [module]
        self.module = self.modules[0]
        self._lane_streams = None
        if test_iter is not None:
            self.data_names = [k[0] for k in test_iter.provide_data_single]
        self.rcnn_output_names = rcnn_output_names or {
            'cls': 'cls_prob_reshape_output', 'bbox': 'bbox_pred_reshape_output', 'im_ids': 'im_ids',
            'scale_map': 'scale_prob_output', 'im_info': 'im_info', 'chip_ids': 'chip_ids'}
        self.rpn_output_names = rpn_output_names or {'scores': 'rois_score', 'rois': 'rois_output', 'im_ids': 'im_ids'}
        self.logger = logger
        self.result_path = getattr(imdb, 'result_path', None)
        self.num_classes = imdb.num_classes
        self.class_names = getattr(imdb, 'classes', None)
        self.num_images = len(roidb)
        self.imdb_name = getattr(imdb, 'name', 'imdb')
        self.nms_worker = nms_worker(cfg.TEST.NMS, cfg.TEST.NMS_SIGMA)
        self.batch_size = batch_size or self.cfg.TEST.BATCH_IMAGES
        self.roidb = roidb
        self.verbose = len(roidb) > 1

    # ---- forward ------------------------------------------------------------------------------
    def forward(self, batch, lane=0):
        mod = self.modules[lane]
        mod.forward(batch, is_train=False)
        return [dict(zip(mod.output_names, i)) for i in zip(*mod.get_outputs(merge_multi_context=False))]

    def get_proposals(self, batch, scales):
        data = dict(zip(self.data_names, batch.data))
        outputs = self.forward(batch)
        scores, rois = [], []
        im_ids = np.array([], dtype=int)
        for gpu_out, gpu_scales in zip(outputs, scales):
            gpu_rois = gpu_out[self.rpn_output_names['rois']].asnumpy()
            gpu_scores = gpu_out[self.rpn_output_names['scores']].asnumpy()
            nper_gpu = gpu_rois.shape[0] // self.batch_size
            im_ids = np.hstack((im_ids, gpu_out[self.rpn_output_names['im_ids']].asnumpy().astype(int)))
            for idx in range(self.batch_size):
                cids = np.where(gpu_rois[:, 0] == idx)[0]
                assert len(cids) == nper_gpu, 'The number of rois per GPU should be fixed!'
                scores.append(gpu_scores[cids])
                rois.append(gpu_rois[cids, 1:] / gpu_scales[idx])
        return scores, rois, data, im_ids

    def detect(self, batch, scales):
        return self._collect(self._launch(batch))

    # The forward pass of batch b + 1 runs on the GPU while the host post-processes batch b (score threshold, per-class lists,
    # border pruning: ~30 ms of numpy per 8-image pass, profiles/r02_infer_profile_after.txt -- serial behind each forward in
    # round 2).  _launch enqueues forward + box decoding + device -> pinned-host copies and returns at once; _collect waits for
    # that batch's copy event only.  A captured forward replays into the SAME output tensors, so the copies are enqueued before
    # the next forward on the same stream and land in one of two alternating pinned sets.
    def _pinned(self, key, like, dtype=None):
        # process-wide and never released: a pinned block returned to torch's host allocator from a garbage-collected Tester
        # makes that allocator query events / free host memory at an arbitrary moment -- inside another executor's hipGraph
        # capture that is an illegal call and the process aborts (seen once, in the -m gpu suite).  One entry per output shape.
        k = (getattr(_SLOT, 'job', 0), key, tuple(like.shape), dtype or like.dtype)
        t = _PINNED.get(k)
        if t is None:
            t = _PINNED[k] = torch.empty(tuple(like.shape), dtype=dtype or like.dtype, pin_memory=True)
        return t

    def _stream(self, lane):
        """The HIP stream of a lane: the caller's current stream for a single lane, one side stream per lane otherwise."""
        if len(self.modules) == 1 or not torch.cuda.is_available():
            return None
        if self._lane_streams is None:
            self._lane_streams = _LANE_STREAMS.setdefault(getattr(_SLOT, 'job', 0), [])
            while len(self._lane_streams) < len(self.modules):
                self._lane_streams.append(torch.cuda.Stream())
        return self._lane_streams[lane]

    def _launch(self, batch, lane=0, compact=None):
        """Enqueue one batch on `lane`.  compact = (cls_thresh, do_pruning): threshold (+ prune) the detections on the GPU
        (sn_det_compact) and bring back rows already grouped by class instead of the raw scores / boxes."""
        st = self._stream(lane)
        if st is None:
            return self._launch_on(batch, lane, compact)
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            return self._launch_on(batch, lane, compact)

    def _launch_on(self, batch, lane, compact):
        if getattr(batch, 'ready_event', None) is not None:   # the prefetch thread's image-preparation launches, on ITS stream
            from .iterators.PrefetchingIter import adopt_batch
            adopt_batch(batch)
        data = dict(zip(self.data_names, batch.data))
        outputs = self.forward(batch, lane)
        flips = self.__dict__.setdefault('_pin_flip', {})
        flip = flips[lane] = 1 - flips.get(lane, 0)
        has_focus_maps = self.rcnn_output_names['scale_map'] in outputs[0]
        parts, parts_dev = [], []
        for g, gpu_out in enumerate(outputs):
            rois = gpu_out[self.rpn_output_names['rois']]._data              # (B*R, 5) device, rows of chip b contiguous
            deltas = gpu_out[self.rcnn_output_names['bbox']]._data           # (B, R, 4)
            infos = gpu_out[self.rcnn_output_names['im_info']]._data         # (B, 3) = h, w, scale
            B, R = int(deltas.shape[0]), int(deltas.shape[1])
            assert rois.shape[0] == B * R, 'The number of rois per GPU should be fixed!'
            boxes = torch.empty((B, R, 4), dtype=torch.float64, device=rois.device)
            hip.call('sn_bbox_decode', rois.contiguous(), deltas.contiguous(), infos.float().contiguous(), boxes, B, R, hip.stream())
            cls = gpu_out[self.rcnn_output_names['cls']]._data
            want = {'im_ids': gpu_out[self.rcnn_output_names['im_ids']]._data,
                    'chip_ids': gpu_out[self.rcnn_output_names['chip_ids']]._data}
            if compact is not None and cls.is_cuda and B <= 64:
                cls_thresh, do_pruning = compact
                NC = int(cls.shape[-1])
                crops = wh = None
                if do_pruning:       # the chips of this batch, from the batch's own (host) id inputs
                    ids = np.asarray(data['im_ids'].asnumpy()).astype(int).reshape(-1)[g * B:(g + 1) * B]
                    cids = np.asarray(data['chip_ids'].asnumpy()).astype(int).reshape(-1)[g * B:(g + 1) * B]
                    crops = np.ascontiguousarray([self.roidb[i]['inference_crops'][c] for i, c in zip(ids, cids)], np.float64)
                    wh = np.ascontiguousarray([[self.roidb[i]['width'], self.roidb[i]['height']] for i in ids], np.float64)
                rows = torch.empty((B, (NC - 1) * R, 5), dtype=torch.float64, device=cls.device)
                counts = torch.empty((B, NC - 1), dtype=torch.int32, device=cls.device)
                hip.call('sn_det_compact', cls.contiguous(), boxes, crops, wh, float(cls_thresh), 10.0, B, R, NC, rows, counts,
                         hip.stream())
                if self.keep_device_rows:
                    ids_b = np.asarray(data['im_ids'].asnumpy()).astype(int).reshape(-1)[g * B:(g + 1) * B]
                    cids_b = np.asarray(data['chip_ids'].asnumpy()).astype(int).reshape(-1)[g * B:(g + 1) * B]
                    dev_parts = [(int(i), int(c), rows[k], counts[k]) for k, (i, c) in enumerate(zip(ids_b, cids_b))]
                else:
                    dev_parts = None
                if self.host_rows or not self.keep_device_rows:
                    want['rows'], want['counts'] = rows, counts
                else:
                    want['device_only'] = torch.zeros((), dtype=torch.int8)          # (host marker: the rows stay in HBM)
                parts_dev.extend(dev_parts or [])
            else:
                want['boxes'], want['cls'] = boxes, cls
            if has_focus_maps:
                want['maps'] = gpu_out[self.rcnn_output_names['scale_map']]._data
            host = {}
            for k, t in want.items():
                if t.is_cuda:
                    host[k] = self._pinned((lane, flip, g, k), t)
                    host[k].copy_(t, non_blocking=True)
                else:
                    host[k] = t
            parts.append((B, host))
        ev = None
        if torch.cuda.is_available():
            ev = torch.cuda.Event()
            ev.record()
        return data, parts, has_focus_maps, ev, parts_dev

    def _collect(self, handle):
        """-> scores, boxes, data, im_ids, maps, chip_ids; after a compacting launch `scores` is None and boxes[i] is the pair
        (rows of chip i grouped by class, float64 (n, 5); rows per class, (NC - 1,))."""
        data, parts, has_focus_maps, ev, parts_dev = handle
        if ev is not None:
            ev.synchronize()
        self._last_device_parts = parts_dev
        scores, preds, maps = [], [], []
        im_ids = np.array([], dtype=int)
        chip_ids = np.array([], dtype=int)
        compacted = False
        for B, host in parts:
            im_ids = np.hstack((im_ids, host['im_ids'].numpy().astype(int)))
            chip_ids = np.hstack((chip_ids, host['chip_ids'].numpy().astype(int)))
            scale_prob = host['maps'].numpy().copy() if has_focus_maps else None     # (the pinned set is reused two batches on)
            if 'rows' in host:
                compacted = True
                counts, rows = host['counts'].numpy().astype(np.int64), host['rows'].numpy()
                for idx in range(B):
                    preds.append((rows[idx, :int(counts[idx].sum())].copy(), counts[idx]))
            elif 'device_only' in host:
                compacted = True
                preds.extend([None] * B)             # thresholded, pruned and kept in HBM (device_parts)
            else:
                gpu_scores, boxes = host['cls'].numpy().copy(), host['boxes'].numpy().copy()
                for idx in range(B):
                    scores.append(gpu_scores[idx])
                    preds.append(boxes[idx])
            if has_focus_maps:
                maps.extend(scale_prob[idx] for idx in range(B))
        return (None if compacted else scores), preds, data, im_ids, maps, chip_ids

    def set_scale(self, scale):
        it = self.test_iter.iters[0] if isinstance(self.test_iter, PrefetchingIter) else self.test_iter
        it.set_scale(scale)
        self.test_iter.reset()

    def show_info(self, print_str):
        print(print_str)
        if self.logger:
            self.logger.info(print_str)

    # ---- multi-scale aggregation (:152-230) ---------------------------------------------------------
    def aggregate(self, scale_cls_dets, vis=False, cache_name='cache', vis_path=None, vis_name=None, pre_nms_db_divide=10,
                  vis_ext='.png'):
        n_scales = len(scale_cls_dets)
        assert n_scales == len(self.cfg.TEST.VALID_RANGES), 'A valid range should be specified for each test scale'
        if not vis:
            done = self.aggregate_device(scale_cls_dets)
            if done is not None:
                return done
        all_boxes = [[[] for _ in range(self.num_images)] for _ in range(self.num_classes)]
        # one batched launch instead of Pool(32).map; the problems go up as the one stacked array they are built as
        rows, sizes = aggregate_problems(scale_cls_dets, self.cfg.TEST.VALID_RANGES, self.num_images, self.num_classes, stacked=True)
        final = self.nms_worker.nms_wrapper.process_stacked(rows, sizes)
        k = 0
        for i in range(self.num_images):
            for j in range(1, self.num_classes):
                all_boxes[j][i] = final[k]
                k += 1
        if self.cfg.TEST.MAX_PER_IMAGE > 0:
            nc = self.num_classes - 1
            for i in range(self.num_images):
                kept = cap_detections_per_image(final[i * nc:(i + 1) * nc], self.cfg.TEST.MAX_PER_IMAGE)
                if kept is not None:
                    for j in range(1, self.num_classes):
                        all_boxes[j][i] = kept[j - 1]
        return all_boxes

    def aggregate_device(self, scale_cls_dets):
        """Tester.aggregate (:152-230) with nothing of it on the host: every chip's thresholded / pruned rows are still in HBM
        (`_Detections.device_parts`, left there by sn_det_compact); the valid-range filter + regrouping per (image, class)
        (sn_aggregate_count / sn_aggregate_scatter), the batched soft-NMS and the MAX_PER_IMAGE rule (sn_det_cap_per_image) run on
        them, and ONE copy brings the final boxes back.  The host's share: the (image, class, part) exclusive scan of a few
        thousand counts.  -> all_boxes, or None when some chip's rows are not on the device / hard NMS is configured (the host
        statement above runs).  Bit-equal to it (tests/test_gpu_inference.py)."""
        nms = self.nms_worker.nms_wrapper
        if nms.thresh > 0 or not torch.cuda.is_available():
            return None
        nc, n_img = self.num_classes - 1, self.num_images
        recs, part_image = [], []
        for i in range(n_img):
            for dets, vr in zip(scale_cls_dets, self.cfg.TEST.VALID_RANGES):
                ready = getattr(dets, 'device_parts', None)
                n_chips = len(dets[1][i]) if len(dets) > 1 else 0
                for c in range(n_chips):
                    hit = ready.get((i, c)) if ready else None
                    if hit is None:
                        return None
                    # `areas > vr * vr` against a float32 array compares in float32 (numpy's weak Python scalars)
                    recs.append((hit[0].data_ptr(), hit[1].data_ptr(), np.float32(vr[0] * vr[0]) if vr[0] > 0 else np.float32(0),
                                 np.float32(vr[1] * vr[1]) if vr[1] > 0 else np.float32(0)))
                    part_image.append(i)
        all_boxes = [[np.zeros((0, 5), np.float32) for _ in range(n_img)] for _ in range(self.num_classes)]
        for i in range(n_img):
            all_boxes[0][i] = []
        P = len(recs)
        if P == 0:
            return all_boxes
        table = np.array(recs, dtype=[('rows', '<u8'), ('counts', '<u8'), ('lo2', '<f4'), ('hi2', '<f4')])      # struct AggPart
        d_table = hip.dev(table.view(np.uint8).reshape(P, 24))
        kept = torch.empty((P, nc), dtype=torch.int32, device=d_table.device)
        CH = 32768                                    # parts per launch (grid.y)
        for a in range(0, P, CH):
            hip.call('sn_aggregate_count', d_table[a:], min(CH, P - a), nc, kept[a:], hip.stream())
        k = kept.cpu().numpy().astype(np.int64)
        part_image = np.asarray(part_image)
        sizes = np.zeros((n_img, nc), np.int64)
        np.add.at(sizes, part_image, k)
        off = np.zeros(n_img * nc + 1, np.int64)
        off[1:] = np.cumsum(sizes.ravel())
        total = int(off[-1])
        if total == 0:
            return all_boxes
        assert total < 2 ** 31
        before = np.cumsum(k, axis=0) - k             # rows of class j in the parts before p (all images)
        first = np.searchsorted(part_image, np.arange(n_img))          # first part of every image (parts are image-major)
        dst = off[:-1].reshape(n_img, nc)[part_image] + before - before[first[part_image]]
        rows = torch.empty((total, 5), dtype=torch.float32, device=d_table.device)
        d_dst = hip.dev(np.ascontiguousarray(dst, np.int32))
        for a in range(0, P, CH):
            hip.call('sn_aggregate_scatter', d_table[a:], min(CH, P - a), nc, d_dst[a:], rows, hip.stream())
        d_off = hip.dev(off.astype(np.int32))
        cnt = torch.empty((n_img * nc,), dtype=torch.int32, device=rows.device)
        from .ext.cpu_nms import _soft_ws
        max_n = int(sizes.max())
        hip.call('sn_soft_nms_batch', rows, d_off, n_img * nc, max_n, total, float(nms.sigma), 0.3, 0.001, 2,
                 _soft_ws(max_n, total, rows.device), cnt, hip.stream())
        if self.cfg.TEST.MAX_PER_IMAGE > 0:
            hip.call('sn_det_cap_per_image', rows, d_off, cnt, n_img, nc, int(self.cfg.TEST.MAX_PER_IMAGE), hip.stream())
        h, c = rows.cpu().numpy(), cnt.cpu().numpy()
        starts, ends = off[:-1].tolist(), (off[:-1] + c).tolist()
        q = 0
        for i in range(n_img):
            for j in range(1, self.num_classes):
                all_boxes[j][i] = h[starts[q]:ends[q]]
                q += 1
        return all_boxes

    # ---- per-scale detection loop (:232-370) --------------------------------------------------------
    @staticmethod
    def _check_valid(det, chip, im_width, im_height, delta=10):
        dx1, dy1, dx2, dy2 = det[0], det[1], det[2], det[3]
        cx1, cy1, cx2, cy2 = chip[0], chip[1], chip[2], chip[3]
        if cx1 >= 0.5 and abs(dx1 - cx1) < delta:
            return False
        if cy1 >= 0.5 and abs(dy1 - cy1) < delta:
            return False
        if cx2 < im_width - 0.5 and abs(dx2 - cx2) < delta:
            return False
        if cy2 < im_height - 0.5 and abs(dy2 - cy2) < delta:
            return False
        return True

    @staticmethod
    def _valid_mask(dets, chip, im_width, im_height, delta=10):
        """_check_valid over the rows of `dets` at once (same comparisons, same float64 arithmetic)."""
        ok = np.ones(len(dets), bool)
        if chip[0] >= 0.5:
            ok &= ~(np.abs(dets[:, 0] - chip[0]) < delta)
        if chip[1] >= 0.5:
            ok &= ~(np.abs(dets[:, 1] - chip[1]) < delta)
        if chip[2] < im_width - 0.5:
            ok &= ~(np.abs(dets[:, 2] - chip[2]) < delta)
        if chip[3] < im_height - 0.5:
            ok &= ~(np.abs(dets[:, 3] - chip[3]) < delta)
        return ok

    def get_detections(self, cls_thresh=1e-3, cache_name='cache', evaluate=False, vis=False, vis_path=None, do_pruning=False,
                       autofocus=False, vis_ext='.png'):
        n_chips = [len(r['inference_crops']) for r in self.roidb]
        all_boxes = _Detections([[[] for _ in range(n_chips[i])] for i in range(self.num_images)] for _ in range(self.num_classes))
        all_maps = [[[] for _ in range(n_chips[i])] for i in range(self.num_images)]
        nc = self.num_classes - 1

        def post(scores, boxes, data, im_ids, maps, chip_ids):
            todo = []
            for i_, c_, rows_, counts_ in getattr(self, '_last_device_parts', None) or ():
                all_boxes.device_parts[(i_, c_)] = (rows_, counts_)
            for i, (cboxes, im_id, chip_id) in enumerate(zip(boxes, im_ids, chip_ids)):
                if autofocus:
                    all_maps[im_id][chip_id] = maps[i]
                if scores is None and cboxes is None:
                    continue                       # the chip's rows stayed in HBM (all_boxes.device_parts)
                if scores is None:
                    # thresholded (and pruned) on the GPU: rows grouped by class + rows per class -- only slice
                    big, lens = cboxes
                    ends = np.cumsum(lens).tolist()
                    start = 0
                    for j in range(nc):
                        all_boxes[j + 1][im_id][chip_id] = big[start:ends[j]]
                        start = ends[j]
                    all_boxes.compact[(im_id, chip_id)] = (big, lens)
                    continue
                # all classes at once: rows grouped by class, RoIs ascending inside a class (= np.where per class, :290-295)
                per_class = threshold_detections(scores[i], cboxes, cls_thresh, self.num_classes)
                for j in range(1, self.num_classes):
                    if evaluate:
                        todo.append((j, im_id, chip_id, per_class[j - 1]))
                    else:
                        all_boxes[j][im_id][chip_id] = per_class[j - 1]
            if scores is None:
                return
            if evaluate:
                final = self.nms_worker.worker_many([t[3] for t in todo])
                for (j, im_id, chip_id, _), d in zip(todo, final):
                    all_boxes[j][im_id][chip_id] = d
                if self.cfg.TEST.MAX_PER_IMAGE:
                    for im_id, chip_id in set((t[1], t[2]) for t in todo):
                        image_scores = np.hstack([all_boxes[j][im_id][chip_id][:, -1] for j in range(1, self.num_classes)])
                        if len(image_scores) > self.cfg.TEST.MAX_PER_IMAGE:
                            image_thresh = np.sort(image_scores)[-self.cfg.TEST.MAX_PER_IMAGE]
                            for j in range(1, self.num_classes):
                                keep = np.where(all_boxes[j][im_id][chip_id][:, -1] >= image_thresh)[0]
                                all_boxes[j][im_id][chip_id] = all_boxes[j][im_id][chip_id][keep, :]
            if do_pruning:     # project the boxes back to image coordinates, drop those cut by a chip border (:336-353)
                for im_id, chip_id in set(zip(im_ids.tolist(), chip_ids.tolist())):
                    crop = self.roidb[im_id]['inference_crops'][chip_id]
                    pruned = prune_chip_border([all_boxes[j][im_id][chip_id] for j in range(1, self.num_classes)], crop,
                                               self.roidb[im_id]['width'], self.roidb[im_id]['height'])
                    for j, d in enumerate(pruned):
                        all_boxes[j + 1][im_id][chip_id] = d
        # Without the per-chip NMS of `evaluate` the threshold and the pruning are one predicate per (RoI, class): the GPU applies
        # it and compacts the rows (sn_det_compact).  Batch k runs on lane k % lanes; a lane's previous batch is collected (in
        # batch order) before the lane is reused, so `lanes` forwards are in flight while the host slices an earlier batch.
        compact = None if (evaluate or not self.device_compact) else (cls_thresh, do_pruning)
        lanes = len(self.modules)
        # one lane (the default): still two batches in flight -- batch k + 1 is enqueued BEFORE batch k is collected, so the host
        # slices k under the forward of k + 1 (a lane's pinned result sets alternate: k is out of its set before k + 2 is launched)
        depth = lanes if lanes > 1 else 2
        pending = []
        for k, batch in enumerate(self.test_iter):
            if len(pending) >= depth:
                post(*self._collect(pending.pop(0)))
            pending.append(self._launch(batch, k % lanes, compact))
        while pending:
            post(*self._collect(pending.pop(0)))
        if self._own_iter:
            self.test_iter.close()               # (the prefetch thread this Tester started)
        return all_boxes, all_maps

    def extract_proposals(self, n_proposals=300, cache_name='cache', vis=False, vis_ext='.png'):
        all_boxes = [[] for _ in range(self.num_images)]
        for batch in self.test_iter:
            im_info = batch.data[1].asnumpy()
            scales = im_info[:, 2].reshape(-1, self.batch_size)
            scores, boxes, data, im_ids = self.get_proposals(batch, scales)
            for cscores, cboxes, im_id in zip(scores, boxes, im_ids):
                all_boxes[im_id] = np.hstack((cboxes[0:n_proposals, 0:4], cscores[0:n_proposals].reshape(-1, 1))).astype(np.float32)
        return all_boxes


def detect_scale_worker(arguments, module_cache=None, lanes=1, image_cache=None, rows=(False, True)):
    """One test scale: bind the test graph for that scale's batch shape and run the Tester (:411-436).
    module_cache (dict, optional): keeps the bound Module of each scale across calls (its executors are cached per
    batch shape), which is what a long-running inference service -- and the throughput benchmark -- wants; the
    reference rebuilds the Module per call.
    lanes: identical bound Modules (same parameters) whose forwards run concurrently on their own HIP streams when the scale
    has more than one batch (Tester.get_detections): a batch of two FocusChips at the finest scale fills half of the 256 CUs
    and most of its ~190 kernels are latency-bound."""
    [scale, scale_i, nbatch, context, config, sym_def, roidb, imdb, arg_params, aux_params, vis] = arguments
    nGPUs = len(context)
    test_iter = MNIteratorTestAutoFocus(roidb=roidb, config=config, batch_size=nGPUs * nbatch, nGPUs=nGPUs, threads=32,
                                        pad_rois_to=400, crop_size=None, test_scale=scale, image_cache=image_cache)
    n_batches = max(1, test_iter.size // max(1, nGPUs * nbatch))
    lanes = max(1, min(int(lanes), n_batches))
    # module_cache[(scale, nbatch)] stays ONE Module (what a caller reading the cache expects); further lanes live under
    # their own key
    mods = []
    if module_cache is not None:
        first = module_cache.get((tuple(scale), nbatch))
        mods = ([first] if first is not None else []) + list(module_cache.get(('__lanes__', tuple(scale), nbatch), []))
    while len(mods) < lanes:
        sym_inst = sym_def(n_proposals=TEST_N_PROPOSALS, test_nbatch=nbatch)
        sym = sym_inst.get_symbol_rcnn(config, is_train=False)
        mod = mx.mod.Module(symbol=sym, context=context, data_names=[k[0] for k in test_iter.provide_data_single], label_names=None)
        mod.slice_inputs = False         # test time: a rank's batches are its own images (imdb_detection_wrapper shards the roidb)
        mod.bind(test_iter.provide_data, test_iter.provide_label, for_training=False)
        if mods:                 # a further lane: the first lane's parameters (a random initialisation must not differ per lane)
            a0, x0 = mods[0].get_params()
            mod.init_params(arg_params=a0, aux_params=x0, allow_missing=False)
        else:
            mod.init_params(arg_params=arg_params, aux_params=aux_params, allow_missing=arg_params is None)
        mods.append(mod)
    if module_cache is not None:
        module_cache[(tuple(scale), nbatch)] = mods[0]
        module_cache[('__lanes__', tuple(scale), nbatch)] = mods[1:]
    tester = Tester(mods[:lanes], imdb, roidb, test_iter, cfg=config, batch_size=nbatch)
    tester.keep_device_rows, tester.host_rows = bool(rows[0]), bool(rows[1])      # (keep in HBM for aggregate_device, copy to the host)
    return tester.get_detections(vis=False, evaluate=False, cache_name='dets_scale_{}x{}'.format(scale[0], scale[1]),
                                 do_pruning=config.TEST.DO_PRUNING[scale_i], autofocus=config.TEST.AUTO_FOCUS)


def default_lanes():
    """Batches of a scale in flight at once when the caller does not say (SNIPER_LANES overrides).  A lane is a bound Module of its
    own -- up to ~12 GB at the finest test scale (activation pool + parameters) -- and three of them are what the throughput numbers
    are quoted on (bench.py, README): the default is 3 on a card with room for them (>= 96 GB free), else 1 (two batches in flight on
    the one stream).  ADVICE r4: the shipped default used to be 1 while the benchmark asked for 3."""
    env = os.environ.get('SNIPER_LANES')
    if env:
        return max(1, int(env))
    try:
        if torch.cuda.is_available():
            free, _ = torch.cuda.mem_get_info()
            return 3 if free >= 96 * (1 << 30) else 1
    except Exception:      # noqa: BLE001 -- no figure: the conservative default
        pass
    return 1


def _rows_to_host(dets, num_classes):
    """The rows of every chip that so far live in HBM only (`device_parts`) -> the host lists and the `compact` form Tester.get_detections
    fills when it copies rows per batch; the device buffers are released.  (The overflow path of the device aggregation's budget.)"""
    nc = num_classes - 1
    for (i, c), (rows, counts) in list(getattr(dets, 'device_parts', {}).items()):
        lens = counts.cpu().numpy().astype(np.int64)
        big = rows[:int(lens.sum())].cpu().numpy()
        ends, start = np.cumsum(lens).tolist(), 0
        for j in range(nc):
            dets[j + 1][i][c] = big[start:ends[j]]
            start = ends[j]
        dets.compact[(i, c)] = (big, lens)
    dets.device_parts = {}


def shard_images(n_images, rank, world):
    """Images of rank `rank`: every world-th image (shard images across ranks).  Interleaved, not contiguous: a roidb is
    sorted by nothing in particular, but its tail may hold the large images."""
    return list(range(int(rank), int(n_images), int(world)))


def merge_rank_detections(gathered, n_images, num_classes, world):
    """Per-rank per-scale detections (what every rank's `_multi_scale_detections` produced for ITS images, in local image order)
    -> per-scale detections over all images in roidb order: image g = rank + world * local.  The counterpart of the reference's
    merge across its forked jobs (lib/inference.py:494-500), done before `aggregate`."""
    n_scales = len(gathered[0])
    merged = []
    for s in range(n_scales):
        d = _Detections([[[] for _ in range(n_images)] for _ in range(num_classes)])
        for r in range(world):
            cls_lists, compact = gathered[r][s]
            ids = shard_images(n_images, r, world)
            for j in range(num_classes):
                for li, g in enumerate(ids):
                    d[j][g] = cls_lists[j][li]
            d.compact.update({(ids[li], c): v for (li, c), v in compact.items()})
        merged.append(d)
    return merged


def imdb_detection_wrapper(sym_def, config, imdb, roidb, context, arg_params, aux_params, vis=False, module_cache=None,
                           focus_map_fn=None, return_scale_dets=False, concurrent_jobs=1, lanes=None, rank=None, world=None,
                           group=None, device_aggregate=None):
    """Multi-scale inference + aggregation (lib/inference.py:439-529), see `_multi_scale_detections`.
    rank / world (default: the initialised torch.distributed group, else one process): rank r runs images r, r + world, ... through
    all the scales -- (image, chip) units are independent, the FocusChips of an image come from its own maps -- the per-scale
    detections are gathered on rank 0 (`gather_object`; rows of a few MB per rank and pass, no tensor collective) and rank 0 alone
    aggregates, as the reference merges its forked jobs before `aggregate` (:494-500).  Other ranks return None.
    device_aggregate (default: on, SNIPER_DEVICE_AGGREGATE=0 turns it off): the thresholded / pruned rows of every chip stay in HBM
    and the valid-range regrouping, the soft-NMS and the MAX_PER_IMAGE rule run on them there (Tester.aggregate_device); the
    rows come to the host per batch only when the per-scale detection lists are asked for (return_scale_dets, vis)."""
    if lanes is None:
        lanes = default_lanes()
    if device_aggregate is None:
        device_aggregate = os.environ.get('SNIPER_DEVICE_AGGREGATE', '1') != '0'
    device_aggregate = bool(device_aggregate) and torch.cuda.is_available() and Tester.device_compact and config.TEST.NMS <= 0
    rows = (device_aggregate, bool(return_scale_dets or vis or not device_aggregate))      # (keep in HBM, copy to the host)
    if world is None:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            world, rank = dist.get_world_size(group), dist.get_rank(group)
        else:
            world, rank = 1, 0
    rank = int(rank or 0)
    if world > 1:
        import torch.distributed as dist
        ids = shard_images(len(roidb), rank, world)
        local = [roidb[g] for g in ids]
        fmap = None if focus_map_fn is None else (lambda s_i, i, c, m: focus_map_fn(s_i, ids[i], c, m))
        dets = _multi_scale_detections(sym_def, config, imdb, local, context, arg_params, aux_params, vis, module_cache, fmap,
                                       concurrent_jobs, lanes, (False, True)) if local else \
            [_Detections([[] for _ in range(imdb.num_classes)]) for _ in config.TEST.SCALES]
        payload = [([list(d[j]) for j in range(imdb.num_classes)], dict(getattr(d, 'compact', {}))) for d in dets]
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(payload, gathered, dst=0, group=group)
        from .engine.executor import resettle_heap
        if rank != 0:
            resettle_heap()
            return (None, None) if return_scale_dets else None
        detections = merge_rank_detections(gathered, len(roidb), imdb.num_classes, world)
    else:
        detections = _multi_scale_detections(sym_def, config, imdb, roidb, context, arg_params, aux_params, vis, module_cache,
                                             focus_map_fn, concurrent_jobs, lanes, rows)
    tester = Tester(None, imdb, roidb, None, cfg=config, batch_size=config.TEST.BATCH_IMAGES[-1])
    out = tester.aggregate(detections, vis=False, cache_name=None)
    # the pass's own objects (iterators, device image cache) are gone; if a new batch shape froze the heap while they were alive,
    # collect what died in cycles and freeze again (engine/executor.py::resettle_heap)
    del tester
    from .engine.executor import resettle_heap
    resettle_heap()
    return (out, detections) if return_scale_dets else out


def _multi_scale_detections(sym_def, config, imdb, roidb, context, arg_params, aux_params, vis=False, module_cache=None,
                            focus_map_fn=None, concurrent_jobs=1, lanes=1, rows=(False, True)):
    """Coarse-to-fine multi-scale inference (:439-529): every image starts as one crop = the whole image; with
    AUTO_FOCUS the FocusPixel maps of scale s generate the chips of scale s+1 (add_chips).  -> the per-scale detections
    (imdb_detection_wrapper aggregates them under TEST.VALID_RANGES with per-class NMS).
    concurrent_jobs (TEST.CONCURRENT_JOBS, :452-500): the roidb is cut into that many contiguous parts and every scale runs the
    parts CONCURRENTLY -- the reference forks one model process per part; here one thread per part, each with its own bound
    Module and its own HIP stream on this process's GPU (a batch of two FocusChips at the finest scale leaves most of the 256
    CUs idle; two streams fill them).  Results are merged in part order, as the reference does (:494-500).
    lanes: batches of one scale in flight at once, each on its own bound Module and HIP stream, driven by ONE host thread
    (detect_scale_worker / Tester.get_detections) -- the form of concurrency that measured faster here than threads.  Every
    lane is a full bound Module (its own activations at every scale, 1400 x 2000 included); default (None): `default_lanes()` -- 3 on a
    card with >= 96 GB free (what bench.py and the README quote), else 1.
    focus_map_fn(scale_i, image, chip, net_map) -> map (benchmarks only): replaces the network's FocusPixel map before the
    FocusChips are cut -- a random-init network's maps select whole images, a trained one's ~10 % of the pixels in blobs
    (SURVEY 8(d)); return_scale_dets: also hand back the per-scale detections (what the CPU baseline of the aggregation reads)."""
    for r in roidb:
        r['inference_crops'] = np.array([[0, 0, r['width'], r['height']]])
    jobs = max(1, min(int(concurrent_jobs), len(roidb)))
    per_job = int(math.ceil(float(len(roidb)) / jobs))
    parts = [roidb[j * per_job:min((j + 1) * per_job, len(roidb))] for j in range(jobs)]
    parts = [p for p in parts if p]
    streams, pool = None, None
    if len(parts) > 1:
        from multiprocessing.pool import ThreadPool
        pool = ThreadPool(len(parts))
        if torch.cuda.is_available():
            streams = module_cache.setdefault('__streams__', [torch.cuda.Stream() for _ in parts]) if module_cache is not None \
                else [torch.cuda.Stream() for _ in parts]
    detections = []
    # one decode + upload per image and PASS, not per scale (data/im_worker.py::DeviceImageCache); SNIPER_IMAGE_CACHE=0: per scale
    image_cache = None
    if os.environ.get('SNIPER_IMAGE_CACHE', '1') != '0' and torch.cuda.is_available():
        from .data.im_worker import DeviceImageCache
        image_cache = DeviceImageCache()
    # rows kept in HBM for the device aggregation are CAPACITY buffers ((classes - 1) x RoIs x 40 B per chip: ~1 MB), whatever
    # survives the threshold: a pass over a few hundred images holds a few GB, a 5000-image roidb at the finest scale would ask
    # for more than the card has.  Budget (SNIPER_DEVICE_AGG_GB, default 24): the scale that would exceed it -- and every later
    # one -- brings its rows to the host per batch as rounds 2-4 did, the earlier scales' rows are brought over once
    # (_rows_to_host), and Tester.aggregate runs its host statement.
    dev_budget = float(os.environ.get('SNIPER_DEVICE_AGG_GB', '24')) * (1 << 30)
    dev_used = 0.0
    for scale_i, (nbatch, scale) in enumerate(zip(config.TEST.BATCH_IMAGES, config.TEST.SCALES)):
        if rows[0]:
            per_chip = (imdb.num_classes - 1) * TEST_N_PROPOSALS * 40.0      # (the RoIs per chip of the graphs bound below: ADVICE r5)
            dev_used += sum(len(r['inference_crops']) for r in roidb) * per_chip
            if dev_used > dev_budget:
                for d in detections:
                    _rows_to_host(d, imdb.num_classes)
                rows = (False, True)

        def job(j):
            was, _SLOT.job = getattr(_SLOT, 'job', 0), j
            try:
                cache = None if module_cache is None else module_cache.setdefault(('__job__', j), {})
                args = [scale, scale_i, nbatch, context, config, sym_def, parts[j], imdb, arg_params, aux_params, vis]
                if streams is None:
                    return detect_scale_worker(args, cache, lanes, image_cache, rows)
                main = torch.cuda.current_stream()
                streams[j].wait_stream(main)
                with torch.cuda.stream(streams[j]):
                    out = detect_scale_worker(args, cache, lanes, image_cache, rows)
                streams[j].synchronize()
                return out
            finally:
                _SLOT.job = was
        if len(parts) == 1:
            dets, maps = detect_scale_worker([scale, scale_i, nbatch, context, config, sym_def, roidb, imdb, arg_params, aux_params, vis],
                                             module_cache, lanes, image_cache, rows)
        else:
            # the first two passes over a module cache run the parts one after the other: that is when the bound executors
            # capture their forward graphs, which must not happen beside another thread's GPU work (engine/executor.py)
            warm = module_cache is None or module_cache.get('__passes__', 0) < 2
            if warm:
                results = [job(j) for j in range(len(parts))]
            else:
                from .engine import executor as _ex
                _ex.set_capture_allowed(False)
                try:
                    results = pool.map(job, range(len(parts)))
                finally:
                    _ex.set_capture_allowed(True)
            dets, maps = results[0]
            for d, m in results[1:]:
                first = len(maps)                # image ids of a part are local to it
                for j in range(imdb.num_classes):
                    dets[j] += d[j]
                maps += m
                dets.compact.update({(first + i, c): v for (i, c), v in getattr(d, 'compact', {}).items()})
                dets.device_parts.update({(first + i, c): v for (i, c), v in getattr(d, 'device_parts', {}).items()})
        detections.append(dets)
        # chips of the next scale from this scale's FocusPixel maps (:497-499)
        if scale_i + 1 < len(config.TEST.SCALES) and config.TEST.DO_PRUNING[scale_i + 1]:
            if focus_map_fn is not None:
                maps = [[focus_map_fn(scale_i, i, j, np.asarray(m)) for j, m in enumerate(mi)] for i, mi in enumerate(maps)]
            add_chips(roidb, maps, scale_i, config)
    if pool is not None:
        pool.close()
    if module_cache is not None:
        module_cache['__passes__'] = module_cache.get('__passes__', 0) + 1
    del image_cache
    return detections
