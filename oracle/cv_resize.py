"""oracle/cv_resize.py -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

``cv2.resize(im, None, None, fx=s, fy=s, interpolation=cv2.INTER_LINEAR)`` on a ``uint8`` (H, W, 3) image, the call
``im_worker.worker`` / ``worker_autofocus`` make (/root/reference/lib/data_utils/data_workers.py:65,107), restated from the
PUBLISHED algorithm of a third-party dependency that is absent from /root/reference and from this image:

    OpenCV (the reference needs ``cv2``; no version is pinned -- README "pip install -r requirements.txt": opencv-python),
    modules/imgproc/src/resize.cpp, identical in the 3.4 and 4.x series for this path:
      cv::resize            dsize = (cvRound(W * fx), cvRound(H * fy)) when dsize is empty; inv_scale = fx, fy kept as given
      cv::hal::resize       scale = 1. / inv_scale (double); INTER_LINEAR with an exact 2 x 2 decimation is replaced by
                            INTER_AREA ("in case of scale_x && scale_y is equal to 2 INTER_AREA (fast) also is equal to
                            INTER_LINEAR"); coefficient tables: fx = (float)((dx + 0.5) * scale_x - 0.5), sx = cvFloor(fx),
                            fx -= sx, left / right border clamps (sx < 0 -> fx = 0, sx = 0; sx >= W - 1 -> fx = 0,
                            sx = W - 1), 8-bit images: ialpha = saturate_cast<short>(coef * INTER_RESIZE_COEF_SCALE), 2048
      HResizeLinear<uchar,int,short,2048>    row[dx] = S[sx] * a0 + S[sx + 1] * a1      (int; S[sx] * 2048 beyond xmax)
      resizeGeneric_Invoker                  the two source rows of dy: clip(sy + k, 0, H), k = 0, 1 (beta is NOT clamped)
      VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>
                                             dst = uchar((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2)
      resizeAreaFast_Invoker + ResizeAreaFastVec (2 x 2, 3 channels)
                                             whole 2 x 2 blocks: (a + b + c + d + 2) >> 2; the last column / row of an odd
                                             source size: saturate_cast<uchar>((float)sum / count) over the pixels that exist
    cvRound / saturate_cast<int>(double|float) round to nearest, ties to even (lrint); cvFloor is floor.
    (IPP builds skip their own 8-bit linear resize unless "not exact" results are allowed -- ipp_resize returns false for
    ipp8u + ippLinear -- so the pip wheels run this same code.)

PINNING: the algorithm is restated from the published source; OpenCV cannot be installed here (no network) and the reference
holds no resized-image fixtures, so no vector minted by cv2 itself backs this file -- "pinned to the published algorithm".
Self-checks in tests/test_oracle_cv_resize.py: scale 1 is the identity, constant images stay constant, the 2 x 2 decimation of
an even-sized image is the rounded block mean, coefficients sum to 2048.
"""
import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS          # INTER_RESIZE_COEF_SCALE


def cv_round(x):
    """cvRound / saturate_cast<int>(double): nearest, ties to even."""
    return int(np.rint(np.float64(x)))


def dsize_of(h, w, fx, fy):
    return cv_round(np.float64(h) * np.float64(fy)), cv_round(np.float64(w) * np.float64(fx))


def _axis_tables(n_dst, n_src, scale):
    """(ofs int32 [n_dst], coef int16 [n_dst, 2], raw float32 frac) of hal::resize's loop over dx (clamp=True) -- the dy loop keeps
    the unclamped offset and coefficient and clips the ROWS instead; both are returned by the two callers below."""
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * np.float64(scale) - 0.5).astype(np.float32)           # (float)((dx+0.5)*scale_x - 0.5)
    s = np.floor(f).astype(np.int32)                                       # cvFloor
    f = (f - s.astype(np.float32)).astype(np.float32)                      # fx -= sx   (float)
    return s, f


def _short_coefs(f):
    c0 = (np.float32(1.0) - f).astype(np.float32) * np.float32(COEF_SCALE)
    c1 = f * np.float32(COEF_SCALE)
    c = np.stack([np.rint(c0), np.rint(c1)], axis=1)                       # saturate_cast<short>(float): cvRound, then clamp
    return np.clip(c, -32768, 32767).astype(np.int32)


def _resize_area_fast_2x2(im, dh, dw):
    H, W, C = im.shape
    src = im.astype(np.int64)
    out = np.zeros((dh, dw, C), np.uint8)
    full_w = W // 2
    for dy in range(dh):
        sy0 = 2 * dy
        if sy0 >= H:
            continue                                                       # rows of zeros (cannot happen for dh = round(H/2))
        w = full_w if sy0 + 2 <= H else 0
        if w:
            blk = src[sy0, 0:2 * w:2] + src[sy0, 1:2 * w:2] + src[sy0 + 1, 0:2 * w:2] + src[sy0 + 1, 1:2 * w:2]
            out[dy, :w] = ((blk + 2) >> 2).astype(np.uint8)
        for dx in range(w, dw):
            sx0 = 2 * dx
            if sx0 >= W:
                continue
            rows = src[sy0:min(sy0 + 2, H), sx0:min(sx0 + 2, W)]
            cnt = rows.shape[0] * rows.shape[1]
            q = (rows.sum(axis=(0, 1)).astype(np.float32) / np.float32(cnt)).astype(np.float32)     # (float)sum / count
            out[dy, dx] = np.clip(np.rint(q), 0, 255).astype(np.uint8)     # saturate_cast<uchar>(float)
    return out


def resize_linear_u8(im, fx, fy=None):
    """cv2.resize(im, None, None, fx=fx, fy=fy, interpolation=cv2.INTER_LINEAR) for uint8 (H, W, C)."""
    fy = fx if fy is None else fy
    im = np.ascontiguousarray(im)
    assert im.dtype == np.uint8 and im.ndim == 3
    H, W, C = im.shape
    dh, dw = dsize_of(H, W, fx, fy)
    assert dh > 0 and dw > 0, 'cv::resize asserts !dsize.empty()'
    scale_x, scale_y = 1.0 / np.float64(fx), 1.0 / np.float64(fy)
    isx, isy = cv_round(scale_x), cv_round(scale_y)
    eps = np.finfo(np.float64).eps
    if abs(scale_x - isx) < eps and abs(scale_y - isy) < eps and isx == 2 and isy == 2:
        return _resize_area_fast_2x2(im, dh, dw)
    # x tables (clamped at both borders)
    sx, ax = _axis_tables(dw, W, scale_x)
    lo = sx < 0
    ax[lo], sx[lo] = 0.0, 0
    hi = sx >= W - 1
    ax[hi], sx[hi] = 0.0, W - 1
    ialpha = _short_coefs(ax)
    # y tables (rows clipped, coefficients as they are)
    sy, ay = _axis_tables(dh, H, scale_y)
    ibeta = _short_coefs(ay)
    src = im.astype(np.int32)
    sx1 = np.minimum(sx + 1, W - 1)                                        # (a1 = 0 wherever sx + 1 would leave the row)
    rows = src[:, sx, :] * ialpha[None, :, 0, None] + src[:, sx1, :] * ialpha[None, :, 1, None]      # (H, dw, C) int
    r0 = np.clip(sy, 0, H - 1)
    r1 = np.clip(sy + 1, 0, H - 1)
    b0, b1 = ibeta[:, 0, None, None], ibeta[:, 1, None, None]
    v = (((b0 * (rows[r0] >> 4)) >> 16) + ((b1 * (rows[r1] >> 4)) >> 16) + 2) >> 2
    return (v & 0xFF).astype(np.uint8)                                     # uchar(int)


def im_prepare(im, crop, scale, flip, means_bgr, out_hw):
    """im_worker.worker / worker_autofocus after cv2.imread (data_workers.py:49-121): optional flip, crop [x1, x2) x [y1, y2)
    clamped to the image, resize, (3, Hm, Wm) float32 with channel j = BGR[2 - j] - PIXEL_MEANS[2 - j] (uint8 - float64 ->
    float64, narrowed to float32 by the assignment), zero padded.  -> (rim, (resized_h, resized_w))."""
    if flip:
        im = im[:, ::-1, :]
    x1, y1, x2, y2 = crop
    im = im[max(int(y1), 0):min(int(y2), im.shape[0]), max(int(x1), 0):min(int(x2), im.shape[1]), :]
    res = resize_linear_u8(im, scale, scale)
    rim = np.zeros((3, out_hw[0], out_hw[1]), np.float32)
    d1m, d2m = min(res.shape[0], out_hw[0]), min(res.shape[1], out_hw[1])
    means = np.asarray(means_bgr, np.float64)
    for j in range(3):
        rim[j, :d1m, :d2m] = res[:d1m, :d2m, 2 - j] - means[2 - j]
    return rim, (res.shape[0], res.shape[1])
