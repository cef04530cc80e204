"""RPN anchor labelling on the GPU, batched over the chips of a minibatch.

Mirror of ``anchor_worker`` (lib/data_utils/data_workers.py:132-371): same constructor inputs
(config + chip size), same per-chip argument list as ``anchor_worker.worker`` --
``[im_info, cur_crop, im_scale, nids, gtids, gt_boxes, boxes, classes]`` -- but all chips of a batch
go through one ``sn_anchor_assign`` call and the outputs are the dense device tensors
``MNIteratorE2E._get_batch`` (lib/iterators/MNIteratorE2E.py:175-194) assembles:
label (B, A*F*F), bbox_target / bbox_weight (B, 4A, F, F), gt_boxes (B, 100, 5).
"""
import numpy as np
import torch

from .. import hip


def generate_anchors(base_size=16, ratios=(0.5, 1, 2), scales=(8, 16, 32)):
    """Base anchors, lib/data_utils/generate_anchor.py:8-77: for every ratio r the (base x base)
    cell is reshaped to w = round(sqrt(base^2 / r)), h = round(w * r), then scaled; all centred on
    the cell centre.  Returns (len(ratios)*len(scales), 4) float64."""
    ratios = np.asarray(ratios, dtype=np.float64)
    scales = np.asarray(scales, dtype=np.float64)
    ctr = 0.5 * (base_size - 1)
    ws = np.round(np.sqrt(float(base_size) * float(base_size) / ratios))
    hs = np.round(ws * ratios)
    w = (ws[:, None] * scales[None, :]).reshape(-1)
    h = (hs[:, None] * scales[None, :]).reshape(-1)
    return np.stack((ctr - 0.5 * (w - 1), ctr - 0.5 * (h - 1), ctr + 0.5 * (w - 1), ctr + 0.5 * (h - 1)), axis=1)


class AnchorAssigner(object):
    MAX_GT = 128

    def __init__(self, cfg, chip_size=512):
        net, tr = cfg.network, cfg.TRAIN
        self.cfg = cfg
        self.feat_stride = int(net.RPN_FEAT_STRIDE)
        self.F = chip_size // self.feat_stride
        self.chip_size = chip_size
        # data_workers.py:134-141: scales go through float32 before generate_anchors
        self.base = generate_anchors(self.feat_stride, list(net.ANCHOR_RATIOS),
                                     list(np.array(net.ANCHOR_SCALES, dtype=np.float32)))
        self.A = self.base.shape[0]
        self.total = self.A * self.F * self.F
        self.rpn_batch = int(tr.RPN_BATCH_SIZE)
        self.num_fg = int(self.rpn_batch * tr.RPN_FG_FRACTION)
        self.pos_thresh = float(tr.RPN_POSITIVE_OVERLAP)
        self.neg_thresh = float(tr.RPN_NEGATIVE_OVERLAP)
        self._ws = hip.Workspace()
        self._d_base = None

    def _pack(self, chips):
        B, G = len(chips), self.MAX_GT
        gt = np.zeros((B, G, 4), np.float32)
        cls = np.zeros((B, G), np.float32)
        inchip = np.zeros((B, G), np.uint8)
        ngt = np.zeros((B,), np.int32)
        crop = np.zeros((B, 2), np.float64)
        scale = np.zeros((B,), np.float32)
        for b, c in enumerate(chips):
            im_info, cur_crop, im_scale, nids, gtids, gt_boxes, boxes, classes = c[:8]
            k = len(gtids)
            if k > G:
                raise ValueError("chip has %d GT boxes; sn_anchor_assign supports at most %d" % (k, G))
            ngt[b] = k
            gt[b, :k] = np.asarray(gt_boxes, np.float32).reshape(-1, 4)
            cls[b, :k] = np.asarray(classes, np.float32).reshape(-1)
            inchip[b, :k] = np.isin(np.asarray(gtids), np.asarray(nids))
            crop[b] = [cur_crop[0], cur_crop[1]]
            scale[b] = np.float32(im_scale)
        return gt, cls, inchip, ngt, crop, scale

    def pack_device(self, chips):
        """The labelling inputs of a chip batch, packed and uploaded once: pass the result as `assign(packed=...)` when
        the same batch is labelled repeatedly (bench.py: inputs resident in HBM; the per-step upload from pageable host
        memory would make the host wait for the previous step)."""
        return (len(chips),) + tuple(hip.dev(a) for a in self._pack(chips))

    def assign(self, chips=None, keys=None, seed=0, want_label_pre=False, packed=None):
        """chips: list of anchor_worker.worker argument lists.  keys: optional (B, A*F*F) uint32
        sub-sampling keys in reference anchor order (see sn_anchor_assign); None = on-device hash."""
        device = hip.require_gpu()
        if packed is None:
            packed = self.pack_device(chips)
        B, gt, cls, inchip, ngt, crop, scale = packed
        if self._d_base is None:
            self._d_base = hip.dev(self.base, torch.float64)
        A, F = self.A, self.F
        out = {
            'label': torch.empty((B, self.total), dtype=torch.float32, device=device),
            'bbox_target': torch.empty((B, 4 * A, F, F), dtype=torch.float32, device=device),
            'bbox_weight': torch.empty((B, 4 * A, F, F), dtype=torch.float32, device=device),
            'gt_boxes': torch.empty((B, 100, 5), dtype=torch.float32, device=device),
            'counts': torch.empty((B, 4), dtype=torch.int32, device=device),
        }
        lp = torch.empty((B, self.total), dtype=torch.int8, device=device) if want_label_pre else None
        ws = self._ws.get(hip.query("sn_anchor_workspace_bytes", B, A, F, self.MAX_GT))
        d_keys = hip.dev(np.ascontiguousarray(keys, np.uint32).view(np.int32), torch.int32) if keys is not None else None
        hip.call("sn_anchor_assign", gt, cls, inchip, ngt, crop, scale,
                 B, self.MAX_GT, self._d_base, A, F, self.feat_stride, self.chip_size, self.chip_size,
                 self.pos_thresh, self.neg_thresh, self.rpn_batch, self.num_fg, d_keys, int(seed), ws,
                 out['label'], out['bbox_target'], out['bbox_weight'], out['gt_boxes'], out['counts'], lp, hip.stream())
        if want_label_pre:
            out['label_pre'] = lp
        return out

    def focus_mask(self, chips):
        """AutoFocus FocusPixel labels of the same chips (gen_mask, data_workers.py:165-192) -> (B, F*F) device fp32."""
        tr = self.cfg.TRAIN
        device = hip.require_gpu()
        gt, cls, inchip, ngt, crop, scale = self._pack(chips)
        B = len(chips)
        out = torch.empty((B, self.F * self.F), dtype=torch.float32, device=device)
        hip.call("sn_focus_mask", hip.dev(gt), hip.dev(ngt), hip.dev(crop), hip.dev(scale), B, self.MAX_GT, self.F, self.feat_stride,
                 self.chip_size, self.chip_size, float(tr.AUTO_FOCUS_DC_LOW), float(tr.AUTO_FOCUS_SMALL_THRESH),
                 float(tr.AUTO_FOCUS_DC_HIGH), out, hip.stream())
        return out

    def numpy_replay_keys(self, label_pre, rng=np.random):
        """Keys that make the device sub-sampling reproduce numpy's draws of
        data_workers.py:327-338 (``npr.choice(fg_inds, ...)`` then ``npr.choice(bg_inds, ...)``),
        consuming `rng` exactly like the reference does, chip by chip."""
        label_pre = np.asarray(label_pre)
        keys = np.zeros(label_pre.shape, np.uint32)
        for b in range(label_pre.shape[0]):
            lp = label_pre[b]
            inside = np.where(lp != -2)[0]          # reference anchor order
            lab = lp[inside].astype(np.float32)
            fg = np.where(lab == 1)[0]
            if len(fg) > self.num_fg:
                dis = rng.choice(fg, size=(len(fg) - self.num_fg), replace=False)
                lab[dis] = -1
                keys[b, inside[dis]] = 1
            num_bg = self.rpn_batch - np.sum(lab == 1)
            bg = np.where(lab == 0)[0]
            if len(bg) > num_bg:
                dis = rng.choice(bg, size=(len(bg) - num_bg), replace=False)
                keys[b, inside[dis]] = 1
        return keys
