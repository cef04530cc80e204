"""Epoch chip database on the GPU: mirror of ``chip_worker`` (lib/data_utils/data_workers.py:374-594).

Same class name, constructor and per-image methods as the reference (``chip_extractor(r)``,
``box_assigner(r)``, ``reset()``), so ``MNIteratorE2E.reset`` can call it unchanged; but the work
is organised for one GPU instead of a 64-process pool: ``extract_batch`` / ``assign_batch`` push every
(image, scale) unit of the roidb through ONE ``sn_chips_generate_batch`` / ``sn_assign_boxes_batch``
launch each (ragged batches), and the per-image methods are the batch of one.

Randomness: the reference shuffles chip candidates with libc ``rand()`` inside cchips.cpp:117.  Here
the order is an explicit permutation per unit drawn from ``numpy.random`` (seedable), or supplied by
the caller through ``perm_fn(unit_key, n_candidates)`` to replay a recorded order bit-exactly.
"""
import numpy as np
import torch

from .. import hip
from ..ext import chips as chips_ext


def _clip_boxes(boxes, im_shape):
    """lib/bbox/bbox_transform.py:35-50 with im_shape = (h, w): x is clipped to w-1, y to h-1."""
    boxes[:, 0::4] = np.maximum(np.minimum(boxes[:, 0::4], im_shape[1] - 1), 0)
    boxes[:, 1::4] = np.maximum(np.minimum(boxes[:, 1::4], im_shape[0] - 1), 0)
    boxes[:, 2::4] = np.maximum(np.minimum(boxes[:, 2::4], im_shape[1] - 1), 0)
    boxes[:, 3::4] = np.maximum(np.minimum(boxes[:, 3::4], im_shape[0] - 1), 0)
    return boxes


class chip_worker(object):
    def __init__(self, cfg, chip_size, perm_fn=None):
        self.valid_ranges = cfg.TRAIN.VALID_RANGES
        self.scales = cfg.TRAIN.SCALES
        self.chip_size = chip_size
        self.use_neg_chips = cfg.TRAIN.USE_NEG_CHIPS
        self.res_based = isinstance(cfg.TRAIN.SCALES[0], (list, tuple))
        self.perm_fn = perm_fn
        self.chip_stride = np.random.randint(56, 60)

    def reset(self):
        self.chip_stride = np.random.randint(56, 60)   # data_workers.py:390-392

    # ---- geometry helpers ----------------------------------------------------------------------
    def _im_scale(self, i, im_size_min, im_size_max):
        """data_workers.py:409-426."""
        spec = self.scales[i]
        if self.res_based:
            lo, hi = spec
            if lo > 0:
                s = float(lo) / float(im_size_min)
                if hi > 0 and np.round(s * im_size_max) > hi:
                    s = float(hi) / float(im_size_max)
            else:
                s = float(hi) / float(im_size_max)
            return s
        return spec / float(im_size_max) if i == len(self.scales) - 1 else spec

    def _perm(self, key, n):
        if self.perm_fn is not None:
            return self.perm_fn(key, n)
        return np.random.permutation(n).astype(np.int32)

    def _generate_units(self, units, keys):
        """units: list of (boxes f32 (n,4) scaled, W, H); returns list of (k,4) float32 chips."""
        prepared, perms = [], []
        for (boxes, W, H), key in zip(units, keys):
            boxes = _clip_boxes(boxes, np.array([H - 1, W - 1]))          # chip_generator.py:24 (sic)
            boxes = np.ascontiguousarray(boxes, dtype=np.float32)
            prepared.append((boxes, W, H, self.chip_size, self.chip_stride))
            perms.append(self._perm(key, chips_ext.num_candidates(W, H, self.chip_size, self.chip_stride))
                         if boxes.shape[0] > 0 else None)
        return chips_ext.generate_batch(prepared, perms)

    # ---- chip extraction (data_workers.py:394-450) ---------------------------------------------
    def extract_batch(self, roidb, keys=None):
        """-> list (per image) of [chip f64(4), im_scale, h, w, scale_idx] lists."""
        n = len(self.scales)
        units, ukeys, meta = [], [], []
        for ii, r in enumerate(roidb):
            width, height = r['width'], r['height']
            smax, smin = max(width, height), min(width, height)
            gt = r['boxes'][np.where(r['max_overlaps'] == 1)[0], :]
            ws = (gt[:, 2] - gt[:, 0]).astype(np.int32)
            hs = (gt[:, 3] - gt[:, 1]).astype(np.int32)
            area = np.sqrt(ws * hs)
            ms = np.maximum(ws, hs)
            for i in range(n):
                s = self._im_scale(i, smin, smax)
                if i == n - 1:
                    ids = np.where(area >= self.valid_ranges[i][0])[0]
                elif i == 0:
                    ids = np.where((area < self.valid_ranges[i][1]) & (ms < (self.chip_size - self.chip_stride - 1) / s) &
                                   (ws >= 2) & (hs >= 2))[0]
                else:
                    ids = np.where((area >= self.valid_ranges[i][0]) & (area < self.valid_ranges[i][1]) &
                                   (ms < (self.chip_size - self.chip_stride - 1) / s))[0]
                units.append((gt[ids, :] * s, int(width * s), int(height * s)))
                ukeys.append((keys[ii] if keys is not None else ii, 'pos', i))
                meta.append((ii, i, s))
        out = self._generate_units(units, ukeys)
        crops = [[] for _ in roidb]
        for (ii, i, s), ch in zip(meta, out):
            r = roidb[ii]
            cur = np.array(ch, dtype=np.float64).reshape(-1, 4) / s
            for chip in cur:
                if i != n - 1:
                    crops[ii].append([chip, s, self.chip_size, self.chip_size, i])
                else:
                    crops[ii].append([chip, s, int(r['height'] * s), int(r['width'] * s), i])
        return crops

    def chip_extractor(self, r):
        return self.extract_batch([r])[0]

    # ---- box assignment (data_workers.py:452-594) ----------------------------------------------
    def _assign(self, unit_chips, unit_boxes, unit_range, unit_mode):
        """Ragged GPU assignment; returns per unit the chip index (or -1) of every box."""
        U = len(unit_chips)
        nb = [len(b) for b in unit_boxes]
        nc = [len(c) for c in unit_chips]
        total = int(sum(nb))
        if total == 0 or sum(nc) == 0:
            return [np.full(k, -1, np.int32) for k in nb]
        chip_off = np.zeros(U + 1, np.int32); chip_off[1:] = np.cumsum(nc)
        box_off = np.zeros(U + 1, np.int32); box_off[1:] = np.cumsum(nb)
        chips = np.concatenate([np.asarray(c, np.float64).reshape(-1, 4) for c in unit_chips], 0)
        boxes = np.concatenate([np.asarray(b, np.float64).reshape(-1, 4) for b in unit_boxes], 0)
        unit_of = np.repeat(np.arange(U, dtype=np.int32), nb)
        out = torch.empty((total,), dtype=torch.int32, device=hip.require_gpu())
        hip.call('sn_assign_boxes_batch', hip.dev(chips), hip.dev(chip_off), hip.dev(boxes), hip.dev(box_off),
                 hip.dev(np.asarray(unit_range, np.float64)), hip.dev(np.asarray(unit_mode, np.int32)), hip.dev(unit_of), U, total,
                 out, hip.stream())
        out = out.cpu().numpy()
        return [out[box_off[u]:box_off[u + 1]] for u in range(U)]

    def assign_batch(self, roidb, keys=None):
        """roidb entries carry 'crops'.  -> list (per image) of (props_in_chips, neg_chips, neg_props)
        (or [props_in_chips] without negative chips), exactly the reference's return value."""
        n = len(self.scales)
        per_im = []
        u_chips, u_boxes, u_range, u_mode, u_meta = [], [], [], [], []
        for ii, r in enumerate(roidb):
            width, height = r['width'], r['height']
            smax, smin = max(width, height), min(width, height)
            widths = (r['boxes'][:, 2] - r['boxes'][:, 0]).astype(np.int32)
            heights = (r['boxes'][:, 3] - r['boxes'][:, 1]).astype(np.int32)
            max_sizes = np.maximum(widths, heights)
            area = np.sqrt(widths * heights)
            cim = [self._im_scale(i, smin, smax) for i in range(n)]
            chips_s = [[] for _ in range(n)]
            ids_s = [[] for _ in range(n)]
            for ci, crop in enumerate(r['crops']):
                chips_s[crop[4]].append(crop[0])
                ids_s[crop[4]].append(ci)
            valid_ids = []
            for si, s in enumerate(cim):
                if si == n - 1:
                    ids = np.where(area >= self.valid_ranges[si][0])[0]
                else:
                    ids = np.where((area < self.valid_ranges[si][1]) & (max_sizes < (self.chip_size - self.chip_stride - 1) / s) &
                                   (widths >= 2) & (heights >= 2))[0]
                valid_ids.append(ids)
            valid_boxes = [r['boxes'][ids].astype(np.float64) for ids in valid_ids]
            per_im.append(dict(cim=cim, ids_s=ids_s, valid_ids=valid_ids, valid_boxes=valid_boxes))
            for si in range(n):
                u_chips.append(np.array(chips_s[si], np.float64).reshape(-1, 4))
                u_boxes.append(valid_boxes[si])
                u_range.append([self.valid_ranges[si][0], self.valid_ranges[si][1]])
                u_mode.append(0 if si == n - 1 else 1)
                u_meta.append((ii, si))
        assigned = self._assign(u_chips, u_boxes, u_range, u_mode)
        results = []
        neg_units, neg_keys, neg_meta = [], [], []
        for ii, r in enumerate(roidb):
            info = per_im[ii]
            props = [[] for _ in range(len(r['crops']))]
            covered = []
            for si in range(n):
                a = assigned[ii * n + si]
                cov = np.zeros(len(info['valid_ids'][si]), dtype=bool)
                for pi in np.where(a >= 0)[0]:
                    props[info['ids_s'][si][a[pi]]].append(info['valid_ids'][si][pi])
                    cov[pi] = True
                covered.append(cov)
            info['props'] = [np.array(p, dtype=np.int32) for p in props]
            info['covered'] = covered
            if self.use_neg_chips:
                for si, s in enumerate(info['cim']):
                    rem = info['valid_boxes'][si][np.where(~covered[si])[0]]
                    neg_units.append((rem * s, int(r['width'] * s), int(r['height'] * s)))
                    neg_keys.append((keys[ii] if keys is not None else ii, 'neg', si))
                    neg_meta.append((ii, si, rem))
        if not self.use_neg_chips:
            return [[info['props']] for info in per_im]
        neg_out = self._generate_units(neg_units, neg_keys)
        n_chips, n_boxes, n_range, n_mode = [], [], [], []
        for (ii, si, rem), ch in zip(neg_meta, neg_out):
            s = per_im[ii]['cim'][si]
            n_chips.append(np.array(ch, dtype=np.float64).reshape(-1, 4) / s)
            n_boxes.append(rem)
            n_range.append([self.valid_ranges[si][0], self.valid_ranges[si][1]])
            n_mode.append(0 if si == n - 1 else 2)
        neg_assigned = self._assign(n_chips, n_boxes, n_range, n_mode)
        for ii, r in enumerate(roidb):
            info = per_im[ii]
            final_chips, final_props = [], []
            for si in range(n):
                u = ii * n + si
                chips_u, a = n_chips[u], neg_assigned[u]
                neg_ids = info['valid_ids'][si][np.where(~info['covered'][si])[0]]
                s = info['cim'][si]
                for c in range(len(chips_u)):
                    mine = neg_ids[np.where(a == c)[0]]
                    if len(mine) > 25 or (len(mine) > 10 and si != 0):
                        final_props.append(np.array(mine, dtype=int))
                        if si != n - 1:
                            final_chips.append([chips_u[c], s, self.chip_size, self.chip_size, si])
                        else:
                            final_chips.append([chips_u[c], s, int(r['height'] * s), int(r['width'] * s), si])
            r['neg_chips'] = final_chips
            r['neg_props_in_chips'] = final_props
            results.append((info['props'], final_chips, final_props))
        return results

    def box_assigner(self, r):
        return self.assign_batch([r])[0]
