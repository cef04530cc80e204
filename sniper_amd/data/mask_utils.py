"""Polygon plumbing of the mask branch (host side): mirrors of `crop_polys` / `poly_encoder`
(lib/data_utils/mask_utils.py:8-46) and of the GT bookkeeping `anchor_worker.worker` does when a chip carries masks
(lib/data_utils/data_workers.py:203-257), producing the `gt_masks` (100, 500) label of a chip.

A roidb entry holds `gt_masks`: one list of polygons per GT object, every polygon a flat [x0, y0, x1, y1, ...] sequence in
image coordinates (lib/dataset/coco.py:244-255).  A chip's label row k describes the k-th GT kept by the RPN labelling
(>= 10 px after shift / scale / round / clip) and is laid out as
    [category (class - 1), n_segments, len(seg 1), ..., len(seg n), seg 1 coordinates ..., seg n coordinates ..., -1 padding]
with as many leading segments as fit `max_poly_len` floats (the count stops at the first segment that does not fit).
Ragged Python lists: this stays on the host (a few hundred floats per chip); the rasterisation is the device's
(sn_mask_rcnn_target)."""
import numpy as np


def crop_polys(polys, crop, im_scale):
    """image coordinates -> chip coordinates: (p - crop origin) * scale, float32 like the reference (:8-19)."""
    ox, oy = crop[0], crop[1]
    out = []
    for obj in polys:
        segs = []
        for seg in obj:
            s = np.array(seg, dtype=np.float32)
            s[0::2] -= ox
            s[1::2] -= oy
            s *= im_scale
            segs.append(s)
        out.append(segs)
    return out


def poly_encoder(polys, cats, max_poly_len=500, max_n_gts=100):
    """(:22-46) -> (max_n_gts, max_poly_len) float32, -1 padded."""
    enc = np.full((max_n_gts, max_poly_len), -1.0, np.float32)
    for i, (obj, cat) in enumerate(zip(polys, cats)):
        if i >= max_n_gts:
            break
        used, lens = 2 + len(obj), []
        for seg in obj:
            if used + len(seg) > max_poly_len:
                break
            used += len(seg)
            lens.append(len(seg))
        row = [np.array([cat, len(lens)], np.float32), np.asarray(lens, np.float32)]
        row += [np.asarray(obj[j], np.float32) for j in range(len(lens))]
        row = np.concatenate(row)
        enc[i, :len(row)] = row
    return enc


def kept_gt(im_info, cur_crop, im_scale, gt_boxes, min_size=10):
    """Indices (into the chip's GT list) of the boxes the RPN labelling keeps, in order: shift by the crop origin, scale,
    round, clip to the chip, drop boxes below `min_size` px (data_workers.py:203-226, float32 arithmetic of numpy
    float32 arrays with Python scalars).  Row k of the device's gt_boxes output is GT kept_gt(...)[k]."""
    gt = np.array(gt_boxes, np.float32, copy=True).reshape(-1, 4)
    gt[:, 0] -= cur_crop[0]
    gt[:, 2] -= cur_crop[0]
    gt[:, 1] -= cur_crop[1]
    gt[:, 3] -= cur_crop[1]
    gt = np.round(gt * np.float32(im_scale))
    gt[:, 0::2] = np.maximum(np.minimum(gt[:, 0::2], im_info[1] - 1), 0)      # clip_boxes, bbox_transform.py:35-50
    gt[:, 1::2] = np.maximum(np.minimum(gt[:, 1::2], im_info[0] - 1), 0)
    ws, hs = gt[:, 2] - gt[:, 0] + 1, gt[:, 3] - gt[:, 1] + 1
    return np.where((ws >= min_size) & (hs >= min_size))[0], gt


def encode_chip_masks(im_info, cur_crop, im_scale, gt_boxes, classes, mask_polys, max_poly_len=500, max_n_gts=100):
    """The `encoded_polys` output of anchor_worker.worker for one chip (data_workers.py:231-257)."""
    ids, _ = kept_gt(im_info, cur_crop, im_scale, gt_boxes)
    if len(ids) == 0:
        return np.full((max_n_gts, max_poly_len), -1.0, np.float32)
    polys = crop_polys(mask_polys, cur_crop, im_scale)
    ids = [i for i in ids if i < len(polys)]
    cls = np.asarray(classes, np.float32).reshape(-1)
    return poly_encoder([polys[i] for i in ids], [cls[i] - 1 for i in ids], max_poly_len, max_n_gts)
