"""oracle/cv_contours.py -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The three OpenCV calls of ``gmask`` (/root/reference/lib/chips/chips_inference.py:12-89: FocusPixel map -> FocusChips), restated from
the PUBLISHED algorithms of a third-party dependency that is absent from /root/reference and from this image (OpenCV, no version
pinned by the reference; modules/imgproc/src/morph.dispatch.cpp, contours.cpp, shapedescr.cpp -- unchanged for these paths across the 3.x /
4.x series):

  cv2.dilate(mask, np.ones((d, d)))     dst(x, y) = max over the kernel's non-zero (x', y') of src(x + x' - ax, y + y' - ay), anchor
                                        (-1, -1) -> the kernel centre (d // 2, d // 2); the default border value of a dilation makes
                                        pixels outside the image not contribute.
  cv2.findContours(m, RETR_LIST, CHAIN_APPROX_NONE | _SIMPLE)
                                        border following of Suzuki & Abe, "Topological structural analysis of digitized binary images
                                        by border following", CVGIP 30 (1985), Algorithm 1, 8-connected 1-components / 4-connected
                                        0-components, on the image framed by one row / column of zeros (since 3.2 findContours copies
                                        the image into such a frame, offset (-1, -1), instead of clearing its border): the raster scan
                                        starts an OUTER border at a 1-pixel whose left neighbour is 0 and a HOLE border at a pixel >= 1
                                        whose right neighbour is 0, follows it marking pixels NBD / -NBD, and RETR_LIST reports every
                                        border, outer and hole alike.  Contours come back newest first (each finished contour is
                                        linked in front of its predecessors: cvInsertNodeIntoTree), i.e. in REVERSE order of their
                                        starting pixels' raster positions.  CHAIN_APPROX_SIMPLE drops collinear interior points of a
                                        contour: its bounding rectangle is that of CHAIN_APPROX_NONE.
  cv2.boundingRect(points)              (min x, min y, max x - min x + 1, max y - min y + 1).

PINNING: restated from the published algorithm; no cv2-minted vector exists ("pinned to the published algorithm").  The product's
FocusChip code (sniper_amd/chips_inference.py, csrc/host_inference.cpp) does not follow borders: it takes the bounding boxes of the
8-connected foreground components and of the enclosed 4-connected background components grown by one cell.
tests/test_oracle_cv_contours.py holds the two against each other -- rectangles and order -- on thousands of random maps, and the
whole of ``gmask`` (dilate, contours, minimum size, paint-and-repeat, scaling) against the product's.
"""
import math

import numpy as np


def dilate_rect(mask, d):
    """cv2.dilate(mask, np.ones((d, d), np.uint8)) for a 2-D array (any dtype; the reference dilates the 0 / 1 float map)."""
    d = int(d)
    m = np.asarray(mask)
    if d <= 1:
        return m.copy()
    H, W = m.shape
    ay = ax = d // 2
    out = np.full((H, W), -np.inf, np.float64)
    for ky in range(d):
        for kx in range(d):
            dy, dx = ky - ay, kx - ax                          # dst(y, x) sees src(y + dy, x + dx)
            ys0, ys1 = max(0, dy), min(H, H + dy)
            xs0, xs1 = max(0, dx), min(W, W + dx)
            if ys0 >= ys1 or xs0 >= xs1:
                continue
            out[ys0 - dy:ys1 - dy, xs0 - dx:xs1 - dx] = np.maximum(out[ys0 - dy:ys1 - dy, xs0 - dx:xs1 - dx], m[ys0:ys1, xs0:xs1])
    return out.astype(m.dtype)


# the 8 neighbours of a pixel in COUNTER-clockwise order as seen on the screen (row index grows downwards), starting east
_NB = ((0, 1), (-1, 1), (-1, 0), (-1, -1), (0, -1), (1, -1), (1, 0), (1, 1))
_IDX = {d: k for k, d in enumerate(_NB)}


def find_contours_list(mask):
    """cv2.findContours(mask, RETR_LIST, CHAIN_APPROX_NONE)[contours]: list of (n, 2) int arrays of (x, y) points, in cv2's order."""
    m = (np.asarray(mask) != 0)
    H, W = m.shape
    f = np.zeros((H + 2, W + 2), np.int32)                    # the zero frame; image pixel (y, x) is f[y + 1, x + 1]
    f[1:-1, 1:-1] = m
    contours = []
    nbd = 1
    for i in range(1, H + 1):
        for j in range(1, W + 1):
            if f[i, j] == 0:
                continue
            if f[i, j] == 1 and f[i, j - 1] == 0:              # (1a) outer border
                start = (0, -1)
            elif f[i, j] >= 1 and f[i, j + 1] == 0:            # (1b) hole border
                start = (0, 1)
            else:
                continue
            nbd += 1
            pts = [(j - 1, i - 1)]
            # (3.1) clockwise from (i2, j2) around (i, j): the first non-zero pixel
            k0 = _IDX[start]
            first = None
            for s in range(8):
                dy, dx = _NB[(k0 - s) % 8]
                if f[i + dy, j + dx] != 0:
                    first = (i + dy, j + dx)
                    break
            if first is None:
                f[i, j] = -nbd                                 # an isolated pixel
                contours.append(np.array(pts, np.int64))
                continue
            i2, j2 = first                                     # (3.2)
            i3, j3 = i, j
            i1, j1 = first
            while True:
                # (3.3) counter-clockwise around (i3, j3), starting from the element AFTER (i2, j2)
                k = _IDX[(i2 - i3, j2 - j3)]
                east_zero_examined = False
                nxt = None
                for s in range(1, 9):
                    dy, dx = _NB[(k + s) % 8]
                    if f[i3 + dy, j3 + dx] != 0:
                        nxt = (i3 + dy, j3 + dx)
                        break
                    if (dy, dx) == (0, 1):
                        east_zero_examined = True
                i4, j4 = nxt
                # (3.4)
                if east_zero_examined:
                    f[i3, j3] = -nbd
                elif f[i3, j3] == 1:
                    f[i3, j3] = nbd
                # (3.5)
                if (i4, j4) == (i, j) and (i3, j3) == (i1, j1):
                    break
                i2, j2 = i3, j3
                i3, j3 = i4, j4
                pts.append((j3 - 1, i3 - 1))
            contours.append(np.array(pts, np.int64))
    contours.reverse()                                         # newest first
    return contours


def bounding_rect(points):
    p = np.asarray(points).reshape(-1, 2)
    x0, y0, x1, y1 = int(p[:, 0].min()), int(p[:, 1].min()), int(p[:, 0].max()), int(p[:, 1].max())
    return x0, y0, x1 - x0 + 1, y1 - y0 + 1


def gmask(mask, d, thresh_value=0.5, ms=16, im_width=0, im_height=0, cscale=1):
    """chips_inference.py:12-89 line by line over the restated calls (Python-2 integer division spelled //)."""
    mask = np.array(mask, np.float32)
    iw = int(math.ceil(float(im_width) / 16))
    ih = int(math.ceil(float(im_height) / 16))
    hot = mask >= thresh_value
    mask[hot] = 1
    mask[~hot] = 0
    mask = dilate_rect(mask, d)
    mask *= 255
    cnts = find_contours_list(mask.astype(np.uint8))

    def place(cnt):
        x, y, w, h = bounding_rect(cnt)
        cx = (x + x + w) // 2
        cy = (y + y + h) // 2
        w = max(ms, w)
        h = max(ms, h)
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
    chips = []
    nchips = -1
    while nchips != len(chips):
        nchips = len(chips)
        chips = []
        for cnt in cnts:
            x, y, w, h = place(cnt)
            mask[y:y + h, x:x + w] = 255
        cnts = find_contours_list(mask.astype(np.uint8))
        for cnt in cnts:
            x, y, w, h = place(cnt)
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
