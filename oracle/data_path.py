"""numpy restatement of the reference's CPU data path (TEST INFRASTRUCTURE ONLY).

Covers SURVEY.md rows a2-a6: chip extraction, box-to-chip assignment (incl. negative-chip mining),
anchor generation and RPN anchor labelling.  Each function cites the reference lines it follows.
Unlike the reference these take explicit parameters instead of a global ``config`` edict, and the
random draws are injectable so that the GPU path can be fed the same randomness.

Pinned against the reference itself (``oracle/ref_py.py``) by ``tests/test_oracle_vs_ref.py`` and
against the golden vectors in ``tests/golden``.
"""
import math

import numpy as np

from . import capi


# --------------------------------------------------------------------------------------------
# box helpers -- lib/bbox/bbox_transform.py
# --------------------------------------------------------------------------------------------
def clip_boxes(boxes, im_shape):
    """lib/bbox/bbox_transform.py:35-50 (in place, returns boxes). im_shape = (h, w)."""
    boxes[:, 0::4] = np.maximum(np.minimum(boxes[:, 0::4], im_shape[1] - 1), 0)
    boxes[:, 1::4] = np.maximum(np.minimum(boxes[:, 1::4], im_shape[0] - 1), 0)
    boxes[:, 2::4] = np.maximum(np.minimum(boxes[:, 2::4], im_shape[1] - 1), 0)
    boxes[:, 3::4] = np.maximum(np.minimum(boxes[:, 3::4], im_shape[0] - 1), 0)
    return boxes


def filter_boxes(boxes, min_size):
    """lib/bbox/bbox_transform.py:52-62."""
    ws = boxes[:, 2] - boxes[:, 0] + 1
    hs = boxes[:, 3] - boxes[:, 1] + 1
    return np.where((ws >= min_size) & (hs >= min_size))[0]


def bbox_transform(ex_rois, gt_rois):
    """nonlinear_transform, lib/bbox/bbox_transform.py:64-90."""
    ew = ex_rois[:, 2] - ex_rois[:, 0] + 1.0
    eh = ex_rois[:, 3] - ex_rois[:, 1] + 1.0
    ecx = ex_rois[:, 0] + 0.5 * (ew - 1.0)
    ecy = ex_rois[:, 1] + 0.5 * (eh - 1.0)
    gw = gt_rois[:, 2] - gt_rois[:, 0] + 1.0
    gh = gt_rois[:, 3] - gt_rois[:, 1] + 1.0
    gcx = gt_rois[:, 0] + 0.5 * (gw - 1.0)
    gcy = gt_rois[:, 1] + 0.5 * (gh - 1.0)
    return np.stack(((gcx - ecx) / (ew + 1e-7), (gcy - ecy) / (eh + 1e-7), np.log(gw / (ew + 1e-7)),
                     np.log(gh / (eh + 1e-7))), axis=1)


def bbox_pred(boxes, deltas):
    """nonlinear_pred, lib/bbox/bbox_transform.py:93-130 (float64)."""
    if boxes.shape[0] == 0:
        return np.zeros((0, deltas.shape[1]))
    boxes = boxes.astype(np.float64, copy=False)
    w = boxes[:, 2] - boxes[:, 0] + 1.0
    h = boxes[:, 3] - boxes[:, 1] + 1.0
    cx = boxes[:, 0] + 0.5 * (w - 1.0)
    cy = boxes[:, 1] + 0.5 * (h - 1.0)
    pcx = deltas[:, 0::4] * w[:, None] + cx[:, None]
    pcy = deltas[:, 1::4] * h[:, None] + cy[:, None]
    pw = np.exp(deltas[:, 2::4]) * w[:, None]
    ph = np.exp(deltas[:, 3::4]) * h[:, None]
    out = np.zeros(deltas.shape)
    out[:, 0::4] = pcx - 0.5 * (pw - 1.0)
    out[:, 1::4] = pcy - 0.5 * (ph - 1.0)
    out[:, 2::4] = pcx + 0.5 * (pw - 1.0)
    out[:, 3::4] = pcy + 0.5 * (ph - 1.0)
    return out


# --------------------------------------------------------------------------------------------
# anchors -- lib/data_utils/generate_anchor.py, lib/data_utils/data_workers.py:143-158
# --------------------------------------------------------------------------------------------
def generate_anchors(base_size=16, ratios=(0.5, 1, 2), scales=(8, 16, 32)):
    """lib/data_utils/generate_anchor.py:8-77: ratio enumeration (rounded w,h) then scale
    enumeration around the centre of the (0,0,base-1,base-1) cell. Returns (R*S, 4) float64,
    ratio-major / scale-minor."""
    ratios = np.asarray(ratios, np.float64)
    scales = np.asarray(scales, np.float64)
    w = h = float(base_size)
    xc = yc = 0.5 * (base_size - 1)
    size_ratios = (w * h) / ratios
    ws = np.round(np.sqrt(size_ratios))
    hs = np.round(ws * ratios)
    out = []
    for rw, rh in zip(ws, hs):
        sw = rw * scales
        sh = rh * scales
        out.append(np.stack((xc - 0.5 * (sw - 1), yc - 0.5 * (sh - 1), xc + 0.5 * (sw - 1), yc + 0.5 * (sh - 1)),
                            axis=1))
    return np.vstack(out)


def all_anchors(feat_stride, ratios, scales, feat_h, feat_w):
    """data_workers.py:143-158: anchors ordered (cell k = y*W + x, anchor a) -> index k*A + a."""
    scales = np.array(scales, dtype=np.float32)  # data_workers.py:135
    base = generate_anchors(base_size=feat_stride, ratios=list(ratios), scales=list(scales))
    sx, sy = np.meshgrid(np.arange(0, feat_w) * feat_stride, np.arange(0, feat_h) * feat_stride)
    shifts = np.vstack((sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel())).transpose()
    A, K = base.shape[0], shifts.shape[0]
    return (base.reshape((1, A, 4)) + shifts.reshape((1, K, 4)).transpose((1, 0, 2))).reshape((K * A, 4))


# --------------------------------------------------------------------------------------------
# chip extraction / box assignment -- lib/data_utils/data_workers.py:394-594
# --------------------------------------------------------------------------------------------
def image_scale(scale_spec, idx, n_scales, im_size_min, im_size_max, res_based):
    """data_workers.py:409-426 (and :473-489)."""
    if res_based:
        min_t, max_t = scale_spec
        if min_t > 0:
            s = float(min_t) / float(im_size_min)
            if max_t > 0 and np.round(s * im_size_max) > max_t:
                s = float(max_t) / float(im_size_max)
        else:
            s = float(max_t) / float(im_size_max)
        return s
    s = scale_spec
    if idx == n_scales - 1:
        s = s / float(im_size_max)
    return s


def _generate(boxes, width, height, chipsize, stride, perm_fn):
    """chip_generator._cgenerate (lib/chips/chip_generator.py:22-26) over the C restatement of
    cchips.cpp.  `perm_fn(n_candidates)` supplies the shuffle (None = identity)."""
    boxes = clip_boxes(boxes, np.array([height - 1, width - 1]))
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    if boxes.shape[0] == 0:
        # cchips.cpp:56-57 returns before touching the RNG
        return np.zeros((0, 4), np.float32)
    ncand = capi.candidate_chips(width, height, chipsize, stride).shape[0]
    perm = perm_fn(ncand) if perm_fn is not None else None
    return capi.chips_generate(boxes, width, height, chipsize, stride, perm)


def chip_extractor(r, scales, valid_ranges, chip_size, chip_stride, perm_fn=None):
    """chip_worker.chip_extractor, data_workers.py:394-450.

    r: roidb entry with 'width','height','boxes' (k,4) float32,'max_overlaps' (k,).
    Returns list of [chip f64(4), im_scale, h, w, scale_idx]."""
    width, height = r['width'], r['height']
    im_size_max, im_size_min = max(width, height), min(width, height)
    res_based = isinstance(scales[0], (list, tuple))
    gt = r['boxes'][np.where(r['max_overlaps'] == 1)[0], :]
    ws = (gt[:, 2] - gt[:, 0]).astype(np.int32)
    hs = (gt[:, 3] - gt[:, 1]).astype(np.int32)
    area = np.sqrt(ws * hs)
    ms = np.maximum(ws, hs)
    n = len(scales)
    out = []
    for i, spec in enumerate(scales):
        im_scale = image_scale(spec, i, n, im_size_min, im_size_max, res_based)
        if i == n - 1:
            ids = np.where(area >= valid_ranges[i][0])[0]
        elif i == 0:
            ids = np.where((area < valid_ranges[i][1]) & (ms < (chip_size - chip_stride - 1) / im_scale) &
                           (ws >= 2) & (hs >= 2))[0]
        else:
            ids = np.where((area >= valid_ranges[i][0]) & (area < valid_ranges[i][1]) &
                           (ms < (chip_size - chip_stride - 1) / im_scale))[0]
        cur = _generate(gt[ids, :] * im_scale, int(width * im_scale), int(height * im_scale), chip_size,
                        chip_stride, perm_fn)
        cur = np.array(cur, dtype=np.float64) / im_scale
        for chip in cur:
            if i != n - 1:
                out.append([chip, im_scale, chip_size, chip_size, i])
            else:
                out.append([chip, im_scale, int(height * im_scale), int(width * im_scale), i])
    return out


def _assign(chips, boxes, box_ids, chip_ids, rng_lo, rng_hi, coarsest, strict_hi, sink, covered):
    """Inner loop shared by the positive (data_workers.py:516-535) and negative (:557-572)
    assignment: each box goes to its argmax-ignore-overlap chip if the clipped intersection is at
    least 1px each way and sqrt(|inter area|) is inside the scale's valid range."""
    ov = capi.ignore_overlaps(chips, boxes)
    max_ids = ov.argmax(axis=0)
    for pi, cid in enumerate(max_ids):
        c, b = chips[cid], boxes[pi]
        x1, x2 = max(c[0], b[0]), min(c[2], b[2])
        y1, y2 = max(c[1], b[1]), min(c[3], b[3])
        area = math.sqrt(abs((x2 - x1) * (y2 - y1)))
        if x2 - x1 >= 1 and y2 - y1 >= 1:
            if coarsest:
                ok = area >= rng_lo
            else:
                ok = area < rng_hi if strict_hi else area <= rng_hi
            if ok:
                sink[chip_ids[cid]].append(box_ids[pi])
                if covered is not None:
                    covered[pi] = True


def box_assigner(r, scales, valid_ranges, chip_size, chip_stride, use_neg_chips=True, perm_fn=None):
    """chip_worker.box_assigner, data_workers.py:452-594.  r additionally carries 'crops'
    (the chip_extractor output).  Returns (props_in_chips, neg_chips, neg_props_in_chips) or
    [props_in_chips] exactly like the reference."""
    width, height = r['width'], r['height']
    im_size_max, im_size_min = max(width, height), min(width, height)
    res_based = isinstance(scales[0], (list, tuple))
    n = len(scales)
    props = [[] for _ in range(len(r['crops']))]
    widths = (r['boxes'][:, 2] - r['boxes'][:, 0]).astype(np.int32)
    heights = (r['boxes'][:, 3] - r['boxes'][:, 1]).astype(np.int32)
    max_sizes = np.maximum(widths, heights)
    area = np.sqrt(widths * heights)
    cim = [image_scale(s, i, n, im_size_min, im_size_max, res_based) for i, s in enumerate(scales)]
    all_chips = [[] for _ in cim]
    all_ids = [[] for _ in cim]
    for ci, crop in enumerate(r['crops']):
        all_chips[crop[4]].append(crop[0])
        all_ids[crop[4]].append(ci)
    all_chips = [np.array(c) for c in all_chips]
    all_ids = [np.array(c) for c in all_ids]
    valid_ids = []
    for si, s in enumerate(cim):
        if si == n - 1:
            ids = np.where(area >= valid_ranges[si][0])[0]
        else:
            ids = np.where((area < valid_ranges[si][1]) & (max_sizes < (chip_size - chip_stride - 1) / s) &
                           (widths >= 2) & (heights >= 2))[0]
        valid_ids.append(ids)
    valid_boxes = [r['boxes'][ids].astype(np.float64) for ids in valid_ids]
    covered = [np.zeros(b.shape[0], dtype=bool) for b in valid_boxes]
    for si, chips in enumerate(all_chips):
        if chips.shape[0] > 0:
            _assign(chips, valid_boxes[si], valid_ids[si], all_ids[si], valid_ranges[si][0], valid_ranges[si][1],
                    si == n - 1, False, props, covered[si])
    if not use_neg_chips:
        return [[np.array(p, dtype=np.int32) for p in props]]
    rem_boxes = [valid_boxes[i][np.where(covered[i] == False)[0]] for i in range(n)]  # noqa: E712
    neg_chips, neg_props, neg_chip_ids = [], [], []
    first = 0
    for si, s in enumerate(cim):
        chips = _generate(rem_boxes[si] * s, int(width * s), int(height * s), chip_size, chip_stride, perm_fn)
        neg_chips.append(np.array(chips, dtype=np.float64).reshape(-1, 4) / s)
        neg_props += [[] for _ in range(len(chips))]
        neg_chip_ids.append(np.arange(first, first + len(chips)))
        first += len(chips)
    neg_ids = [valid_ids[i][np.where(covered[i] == False)[0]] for i in range(n)]  # noqa: E712
    for si in range(n):
        if neg_chips[si].shape[0] > 0:
            _assign(neg_chips[si], rem_boxes[si], neg_ids[si], neg_chip_ids[si], valid_ranges[si][0],
                    valid_ranges[si][1], si == n - 1, True, neg_props, None)
    final_chips, final_props = [], []
    counter = 0
    for si, chips in enumerate(neg_chips):
        for chip in chips:
            k = len(neg_props[counter])
            if k > 25 or (k > 10 and si != 0):
                final_props.append(np.array(neg_props[counter], dtype=int))
                if si != n - 1:
                    final_chips.append([chip, cim[si], chip_size, chip_size, si])
                else:
                    final_chips.append([chip, cim[si], int(height * cim[si]), int(width * cim[si]), si])
            counter += 1
    return [np.array(p, dtype=np.int32) for p in props], final_chips, final_props


# --------------------------------------------------------------------------------------------
# RPN anchor labelling -- lib/data_utils/data_workers.py:133-371
# --------------------------------------------------------------------------------------------
class AnchorTarget(object):
    """anchor_worker (data_workers.py:132-162 for the setup, :164-371 for the per-chip work)."""

    def __init__(self, chip_size=512, feat_stride=16, ratios=(0.5, 1, 2), scales=(2, 4, 7, 10, 13, 16, 24),
                 rpn_batch=256, fg_fraction=0.5, pos_thresh=0.5, neg_thresh=0.4, max_gts=100,
                 auto_focus=False, af_dc_low=-1, af_dc_high=-1, af_small=-1):
        self.feat_stride = feat_stride
        self.F = chip_size // feat_stride
        self.anchors = all_anchors(feat_stride, ratios, scales, self.F, self.F)
        self.A = self.anchors.shape[0] // (self.F * self.F)
        self.batch = rpn_batch
        self.num_fg = int(rpn_batch * fg_fraction)
        self.pos_thresh, self.neg_thresh = pos_thresh, neg_thresh
        self.max_gts = max_gts
        self.auto_focus = auto_focus
        self.af = (af_dc_low, af_dc_high, af_small)

    def focus_mask(self, gt_boxes):
        """gen_mask, data_workers.py:165-192."""
        F, fs = self.F, self.feat_stride
        lo, hi, small = self.af
        m = np.zeros((F, F), np.float32)
        for b in gt_boxes:
            area = np.sqrt((b[2] - b[0]) * (b[3] - b[1]))
            x1, y1 = int(b[0] / fs), int(b[1] / fs)
            x2, y2 = int(math.ceil(b[2] / fs)), int(math.ceil(b[3] / fs))
            flag = 0
            if area > lo and area < small:
                flag = 1
            elif area >= small and area < hi:
                flag = -1
            elif area <= lo:
                flag = -1
            if flag != 0:
                m[y1:min(y2 + 1, F), x1:min(x2 + 1, F)] = float(flag)
        return m.reshape(F * F)

    def prepare_boxes(self, im_info, cur_crop, im_scale, nids, gtids, gt_boxes, boxes, classes):
        """data_workers.py:203-280: shift into the chip, scale, round, clip, drop <10px boxes, split
        GT into valid (IoU==1 with a box assigned to this chip) and invalid.  float32 arithmetic
        exactly as numpy does it for float32 arrays op python floats."""
        gt = np.array(gt_boxes, np.float32, copy=True)
        vgt = np.array(boxes, np.float32, copy=True)[np.intersect1d(gtids, nids)]
        for arr in (gt, vgt):
            arr[:, 0] -= cur_crop[0]
            arr[:, 2] -= cur_crop[0]
            arr[:, 1] -= cur_crop[1]
            arr[:, 3] -= cur_crop[1]
        gt = clip_boxes(np.round(gt * np.float32(im_scale)), im_info[:2])
        mask = self.focus_mask(gt) if self.auto_focus else None
        vgt = clip_boxes(np.round(vgt * np.float32(im_scale)), im_info[:2])
        classes = np.asarray(classes).reshape(-1, 1)
        ids = filter_boxes(gt, 10)
        if len(ids) == 0:
            gt = np.zeros((0, 4))
            classes = np.zeros((0, 1))
        else:
            gt = gt[ids]
            classes = classes[ids]
        agt = gt.copy()
        ids = filter_boxes(vgt, 10)
        vgt = vgt[ids] if len(ids) > 0 else np.zeros((0, 4))
        if len(vgt) > 0 and len(gt) > 0:
            mov = capi.bbox_overlaps(gt, vgt).max(axis=1)
        else:
            mov = np.zeros((len(gt)))
        invalid = gt[np.where(mov < 1)[0], :]
        valid = gt[np.where(mov == 1)[0], :]
        return valid, invalid, agt, classes, mask

    def label_anchors(self, im_info, valid, invalid):
        """data_workers.py:196-201, 295-325: labels before subsampling, on the inside anchors."""
        a = self.anchors
        inside = np.where((a[:, 0] >= -32) & (a[:, 1] >= -32) & (a[:, 2] < im_info[0] + 32) &
                          (a[:, 3] < im_info[1] + 32))[0]
        anchors = a[inside, :]
        labels = np.full((len(inside),), -1, np.float32)
        argmax = None
        maxn = None
        if len(invalid) > 0:
            maxn = capi.bbox_overlaps(anchors, invalid.astype(np.float64)).max(axis=1)
        if valid.size > 0:
            ov = capi.bbox_overlaps(anchors, valid.astype(np.float64))
            argmax = ov.argmax(axis=1)
            mx = ov[np.arange(len(inside)), argmax]
            gt_max = ov.max(axis=0)
            gt_arg = np.where(ov == gt_max)[0]
            labels[mx < self.neg_thresh] = 0
            labels[gt_arg] = 1
            labels[mx >= self.pos_thresh] = 1
        else:
            labels[:] = 0
        if maxn is not None:
            labels[maxn > 0.3] = -1
        return inside, anchors, labels, argmax

    def subsample(self, labels, choice):
        """data_workers.py:327-338.  choice(inds, size) -> the indices to disable."""
        fg = np.where(labels == 1)[0]
        if len(fg) > self.num_fg:
            labels[choice(fg, len(fg) - self.num_fg)] = -1
        num_bg = self.batch - np.sum(labels == 1)
        bg = np.where(labels == 0)[0]
        if len(bg) > num_bg:
            labels[choice(bg, int(len(bg) - num_bg))] = -1
        return labels

    def __call__(self, im_info, cur_crop, im_scale, nids, gtids, gt_boxes, boxes, classes, choice=None):
        """Dense equivalent of anchor_worker.worker + the scatter in MNIteratorE2E._get_batch
        (lib/iterators/MNIteratorE2E.py:186-194): returns
          label (A*F*F,) float32 holding the float16-rounded labels, layout (a, y, x);
          bbox_target, bbox_weight (4A, F, F) float32; gt (100,5) float32; [mask (F*F)]."""
        if choice is None:
            choice = lambda inds, size: np.random.choice(inds, size=size, replace=False)  # noqa: E731
        valid, invalid, agt, classes, mask = self.prepare_boxes(im_info, cur_crop, im_scale, nids, gtids, gt_boxes,
                                                                boxes, classes)
        inside, anchors, labels, argmax = self.label_anchors(im_info, valid, invalid)
        labels = self.subsample(labels, choice)
        n_in = len(inside)
        targets = np.zeros((n_in, 4), np.float32)
        if valid.size > 0:
            targets[:] = bbox_transform(anchors, valid[argmax, :4])
        weights = np.zeros((n_in, 4), np.float32)
        weights[labels == 1, :] = 1.0
        A, F = self.A, self.F
        total = A * F * F
        lab_all = np.full((total,), -1, np.float32)
        lab_all[inside] = labels
        tgt_all = np.zeros((total, 4), np.float32)
        tgt_all[inside] = targets
        w_all = np.zeros((total, 4), np.float32)
        w_all[inside] = weights
        lab_out = lab_all.reshape((F, F, A)).transpose(2, 0, 1).reshape(total).astype(np.float16).astype(np.float32)
        tgt_t = tgt_all.reshape((F, F, A * 4)).transpose(2, 0, 1)
        w_t = w_all.reshape((F, F, A * 4)).transpose(2, 0, 1)
        # the reference ships (values, pids) and the iterator scatters them into zeros: only
        # positions with weight==1 carry a target
        tgt_out = np.where(w_t == 1, tgt_t, 0).astype(np.float32)
        w_out = np.ascontiguousarray(w_t, np.float32)
        fgt = -np.ones((self.max_gts, 5), np.float32)
        k = min(len(agt), self.max_gts)
        if k > 0:
            fgt[:k, :4] = agt[:k]
            fgt[:k, 4] = np.asarray(classes, np.float64).reshape(-1)[:k]
        out = [lab_out, tgt_out, w_out, fgt]
        if self.auto_focus:
            out.append(mask)
        return out
