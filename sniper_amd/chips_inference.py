"""AutoFocus FocusChip generation, contract of lib/chips/chips_inference.py:12-173 (`gmask`, `add_chips`).

The reference does this on the host with OpenCV (threshold -> cv2.dilate -> cv2.findContours(RETR_LIST) ->
cv2.boundingRect, iterated until the chip count is stable); the maps are tiny (<= 125 x 88 cells).  OpenCV is not part
of this image.  `gmask` runs sn_focus_chips_host (csrc/host_inference.cpp), which follows borders like OpenCV's contours.cpp
(Suzuki & Abe 1985) and returns the chips in cv2's order (newest contour first); it is held -- rectangles and order -- against
oracle/cv_contours.py, the restatement of the three cv2 calls from their published algorithms (tests/test_oracle_cv_contours.py).
`gmask_reference` below is the SECOND route to the same rectangles, with scipy.ndimage (raster order of the components, holes last):

  * cv2.dilate(mask, ones(d, d)) : anchor at the kernel centre (d // 2, d // 2), borders ignored
        -> ndimage.maximum_filter(size=d, mode='constant', cval=0) (same window for odd and even d);
  * cv2.findContours(RETR_LIST) + boundingRect: one contour per 8-connected foreground component (outer border:
        bounding box of the component) and one per hole = 4-connected background component that does not touch the
        image border (hole border = the foreground pixels around it: the hole's bounding box grown by one cell).
Everything else (minimum side `ms`, clamping to the map, paint-and-repeat, x16, clamp to the crop, / cscale) follows the
reference line by line, with Python-2 integer division made explicit."""
import math

import numpy as np
from scipy import ndimage

_S8 = np.ones((3, 3), bool)


def _dilate(mask, d):
    if d <= 1:
        return mask.copy()
    # cv2.dilate anchors the d x d kernel at (d // 2, d // 2): window rows y - d//2 .. y + d - 1 - d//2, which is
    # scipy's default window for both parities; pixels outside the image do not contribute
    return ndimage.maximum_filter(mask, size=d, mode='constant', cval=0)


def _bounding_rects(mask):
    """(x, y, w, h) of every contour cv2.findContours(RETR_LIST) would return on a 0/255 mask."""
    rects = []
    fg = mask > 0
    lab, n = ndimage.label(fg, structure=_S8)
    for sl in ndimage.find_objects(lab):
        rects.append((sl[1].start, sl[0].start, sl[1].stop - sl[1].start, sl[0].stop - sl[0].start))
    H, W = mask.shape
    blab, bn = ndimage.label(~fg)                       # holes: 4-connected background
    for k, sl in enumerate(ndimage.find_objects(blab), 1):
        if sl[0].start == 0 or sl[1].start == 0 or sl[0].stop == H or sl[1].stop == W:
            continue                                     # open background, not a hole
        rects.append((sl[1].start - 1, sl[0].start - 1, sl[1].stop - sl[1].start + 2, sl[0].stop - sl[0].start + 2))
    return rects


def _place(rect, ms, iw, ih):
    x, y, w, h = rect
    cx, cy = (x + x + w) // 2, (y + y + h) // 2
    w, h = max(ms, w), max(ms, h)
    if cx + w // 2 >= iw:
        x = iw - w if iw - w >= 0 else 0
    elif cx - w // 2 < 0:
        x = 0
    else:
        x = cx - w // 2
    if cy + h // 2 >= ih:
        y = ih - h if ih - h >= 0 else 0
    elif cy - h // 2 < 0:
        y = 0
    else:
        y = cy - h // 2
    return x, y, w, h


def gmask(mask, d, thresh_value=0.5, ms=16, im_width=0, im_height=0, cscale=1):
    """The FocusChips of one FocusPixel map: sn_focus_chips_host (csrc/host_inference.cpp), the native form of gmask_reference below
    (same chips; the native code reports them in cv2's order; tests/test_focus_chips.py holds the two against each other).  250 us -> ~10 us per
    map: FocusChip generation sits between two scales of a test pass, where nothing overlaps it."""
    from . import hip
    m = np.ascontiguousarray(mask, np.float32)
    H, W = m.shape
    out, n = np.empty((1024, 4), np.float64), np.zeros(1, np.int32)
    hip.call('sn_focus_chips_host', m, H, W, int(d), float(thresh_value), int(ms), float(im_width), float(im_height), float(cscale),
             out, out.shape[0], n)
    return out[:int(n[0])].tolist()


def gmask_reference(mask, d, thresh_value=0.5, ms=16, im_width=0, im_height=0, cscale=1):
    iw, ih = int(math.ceil(float(im_width) / 16)), int(math.ceil(float(im_height) / 16))
    m = (np.asarray(mask) >= thresh_value).astype(np.uint8)
    m = _dilate(m, int(d)) * np.uint8(255)
    rects = _bounding_rects(m)
    chips, nchips = [], -1
    while nchips != len(chips):
        nchips = len(chips)
        chips = []
        for r in rects:
            x, y, w, h = _place(r, ms, iw, ih)
            m[y:y + h, x:x + w] = 255
        rects = _bounding_rects(m)
        for r in rects:
            x, y, w, h = _place(r, ms, iw, ih)
            chips.append([x, y, x + w, y + h])
    schips = []
    for c in chips:
        x1, y1, x2, y2 = c[0] * 16, c[1] * 16, c[2] * 16, c[3] * 16
        if x2 > im_width:
            x2 = im_width
            x1 = max(min(x1, x2 - ms * 16), 0)
        if y2 > im_height:
            y2 = im_height
            y1 = max(min(y1, y2 - ms * 16), 0)
        schips.append([x1 / cscale, y1 / cscale, x2 / cscale, y2 / cscale])
    return schips


def add_chips(roidb, maps, scale_id, cfg):
    """FocusChips of test scale scale_id+1 from the FocusPixel maps of scale scale_id (reference :91-173).
    maps[i][j] is the (2, h, w) scale_prob of chip j of image i; roidb[i]['inference_crops'] is replaced."""
    from .data.im_worker import target_scale
    total_area = chip_area = 0.0
    for i, r in enumerate(roidb):
        cur_chips = []
        im_width, im_height = r['width'], r['height']
        cscale = target_scale(im_width, im_height, cfg.TEST.SCALES[scale_id])
        tcscale = target_scale(im_width, im_height, cfg.TEST.SCALES[scale_id + 1])
        total_area += (im_width * im_height * tcscale * tcscale) / (1000. * 1000.)
        for j in range(len(maps[i])):
            cmap = maps[i][j][1]
            cur_crop = r['inference_crops'][j]
            crop_width, crop_height = cur_crop[2] - cur_crop[0], cur_crop[3] - cur_crop[1]
            d, thr, ms = cfg.TEST.CHIP_HYPERPARAMS[scale_id]
            chips = gmask(cmap, d, thr, ms=ms, im_width=crop_width * cscale, im_height=crop_height * cscale, cscale=cscale)
            for c in chips:
                c[0] += cur_crop[0]; c[1] += cur_crop[1]; c[2] += cur_crop[0]; c[3] += cur_crop[1]
                chip_area += ((c[2] - c[0]) * (c[3] - c[1]) * tcscale * tcscale) / (1000. * 1000.)
            cur_chips += chips
        roidb[i]['inference_crops'] = np.array(cur_chips)
    return [chip_area, total_area]
