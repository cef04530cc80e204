"""Seeded synthetic COCO-shaped inputs.

There is no dataset in the benchmark environment, so every measurement and parity test runs on a
synthetic roidb with the reference's dict schema (``lib/dataset/coco.py:234-243`` +
``lib/data_utils/load_data.py:33-35``): per image ``width``, ``height``, ``boxes`` (k,4) float32
that went through uint16 rounding, ``gt_classes``, ``max_classes``, ``max_overlaps`` (1 for GT
rows, proposal rows get their max IoU with a GT), ``flipped``.
"""
import numpy as np

_SIZES = [((640, 480), 0.45), ((480, 640), 0.15), ((640, 427), 0.20), ((500, 375), 0.10), ((427, 640), 0.10)]


def _iou_max(boxes, gts):
    if len(gts) == 0 or len(boxes) == 0:
        return np.zeros(len(boxes), np.float32)
    x1 = np.maximum(boxes[:, None, 0], gts[None, :, 0])
    y1 = np.maximum(boxes[:, None, 1], gts[None, :, 1])
    x2 = np.minimum(boxes[:, None, 2], gts[None, :, 2])
    y2 = np.minimum(boxes[:, None, 3], gts[None, :, 3])
    iw = np.clip(x2 - x1 + 1, 0, None)
    ih = np.clip(y2 - y1 + 1, 0, None)
    inter = iw * ih
    a = (boxes[:, 2] - boxes[:, 0] + 1) * (boxes[:, 3] - boxes[:, 1] + 1)
    b = (gts[:, 2] - gts[:, 0] + 1) * (gts[:, 3] - gts[:, 1] + 1)
    return (inter / (a[:, None] + b[None, :] - inter)).max(axis=1).astype(np.float32)


def make_polygons(rs, box, long_prob=0.03):
    """COCO-style segmentation of one object (lib/dataset/coco.py:244-255): 1-3 polygons, each a flat
    [x0, y0, x1, y1, ...] list of a noisy ellipse inscribed in (a part of) the box; now and then a polygon with hundreds of
    vertices, so that the 500-float budget of `poly_encoder` truncates."""
    x1, y1, x2, y2 = [float(v) for v in box]
    segs = []
    for _ in range(int(rs.randint(1, 4))):
        nv = int(rs.randint(130, 300)) if rs.uniform() < long_prob else int(rs.randint(3, 13))
        cx, cy = rs.uniform(x1 + 0.3 * (x2 - x1), x2 - 0.3 * (x2 - x1)), rs.uniform(y1 + 0.3 * (y2 - y1), y2 - 0.3 * (y2 - y1))
        rx, ry = min(cx - x1, x2 - cx), min(cy - y1, y2 - cy)
        ang = np.sort(rs.uniform(0, 2 * np.pi, nv))
        rad = rs.uniform(0.6, 1.0, nv)
        px, py = cx + rx * rad * np.cos(ang), cy + ry * rad * np.sin(ang)
        segs.append([float(v) for v in np.round(np.stack((px, py), 1).reshape(-1), 2)])
    return segs


def make_roidb(n_images=5000, seed=0, n_proposals=0, num_classes=81, with_masks=False):
    """Returns list[dict] (the reference's roidb).  GT rows first, then `n_proposals` uniform boxes
    per image (the merged layout of lib/dataset/imdb.py:398-419).  with_masks adds `gt_masks` (drawn from a separate
    stream: the boxes of a seed do not depend on it)."""
    rs = np.random.RandomState(seed)
    sizes = [s for s, _ in _SIZES]
    probs = np.array([p for _, p in _SIZES])
    roidb = []
    for i in range(n_images):
        W, H = sizes[rs.choice(len(sizes), p=probs)]
        k = int(np.clip(1 + rs.poisson(6.3), 1, 100))
        side = np.exp(rs.uniform(np.log(8.0), np.log(0.8 * min(W, H)), size=k))
        asp = np.exp(rs.normal(0.0, 0.5, size=k))
        bw = np.clip(side * np.sqrt(asp), 2, W - 2)
        bh = np.clip(side / np.sqrt(asp), 2, H - 2)
        x1 = rs.uniform(0, W - 1 - bw)
        y1 = rs.uniform(0, H - 1 - bh)
        gt = np.stack((x1, y1, x1 + bw, y1 + bh), axis=1)
        gt = np.round(gt).astype(np.uint16).astype(np.float32)
        cls = rs.randint(1, num_classes, size=k).astype(np.int32)
        boxes, max_ov, max_cls = gt, np.ones(k, np.float32), cls.copy()
        if n_proposals > 0:
            pw = rs.uniform(8, 0.6 * W, size=n_proposals)
            ph = rs.uniform(8, 0.6 * H, size=n_proposals)
            px = rs.uniform(0, W - 1 - pw)
            py = rs.uniform(0, H - 1 - ph)
            props = np.round(np.stack((px, py, px + pw, py + ph), axis=1)).astype(np.uint16).astype(np.float32)
            pov = _iou_max(props, gt)
            pov[pov >= 1.0] = 0.999  # only GT rows carry max_overlaps == 1 (data_workers.py:400)
            boxes = np.vstack((gt, props))
            max_ov = np.concatenate((max_ov, pov))
            max_cls = np.concatenate((max_cls, np.zeros(n_proposals, np.int32)))
        extra = {}
        if with_masks:
            prs = np.random.RandomState([seed, i, 7])
            extra['gt_masks'] = [make_polygons(prs, b) for b in gt]
        roidb.append({
            'image': 'synthetic_%06d.jpg' % i,
            'width': int(W),
            'height': int(H),
            'boxes': boxes,
            'gt_classes': np.concatenate((cls, np.zeros(len(boxes) - k, np.int32))),
            'max_classes': max_cls,
            'max_overlaps': max_ov,
            'flipped': False,
            **extra,
        })
    return roidb


def make_chips(batch, seed=0, size=512, device=None):
    """Training chip pixels: N(0, 50^2) float32 (B,3,size,size), the mean-subtracted range the
    reference's im_worker produces (data_workers.py:111-116)."""
    rs = np.random.RandomState(seed)
    return (rs.standard_normal((batch, 3, size, size)) * 50.0).astype(np.float32)
