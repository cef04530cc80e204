"""oracle/cv_resize.py (numpy, vectorised) and sniper_oracle.c::orc_cv_resize_linear_u8c3 (scalar, OpenCV's loop structure) are two
independent restatements of cv2.resize(uint8, fx, fy, INTER_LINEAR): they must agree bit for bit, and both must have the properties
the published fixed-point algorithm has.  (No cv2 in this image: "pinned to the published algorithm", see oracle/cv_resize.py.)"""
import numpy as np
import pytest

from oracle import capi, cv_resize

SCALES = [1.0, 0.5, 3.0, 512.0 / 480.0, 1.0 / 0.6, 2.9167, 0.8, 1400.0 / 480.0, 0.25, 1.0 / 3.0, 2.0, 0.49999, 0.5000001, 7.3, 0.07]


@pytest.mark.parametrize('scale', SCALES)
def test_numpy_and_scalar_c_restatements_agree(scale):
    rs = np.random.RandomState(int(scale * 1000) % 9973)
    for (h, w) in ((37, 53), (1, 1), (2, 2), (1, 17), (16, 1), (48, 64), (15, 15), (7, 10), (33, 32)):
        if cv_resize.dsize_of(h, w, scale, scale)[0] < 1 or cv_resize.dsize_of(h, w, scale, scale)[1] < 1:
            continue
        im = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        a = cv_resize.resize_linear_u8(im, scale)
        b = capi.cv_resize_linear_u8c3(im, scale)
        assert a.shape == b.shape == cv_resize.dsize_of(h, w, scale, scale) + (3,)
        assert np.array_equal(a, b), (scale, h, w, np.abs(a.astype(int) - b.astype(int)).max())


def test_anisotropic_scales_agree():
    rs = np.random.RandomState(5)
    im = rs.randint(0, 256, (29, 41, 3)).astype(np.uint8)
    for fx, fy in ((1.3, 0.7), (0.5, 2.0), (2.0, 0.5), (0.5, 0.5000001)):
        assert np.array_equal(cv_resize.resize_linear_u8(im, fx, fy), capi.cv_resize_linear_u8c3(im, fx, fy))


def test_properties_of_the_fixed_point_algorithm():
    rs = np.random.RandomState(1)
    im = rs.randint(0, 256, (24, 36, 3)).astype(np.uint8)
    # scale 1: coefficients (2048, 0) on both axes -> (2048 * (S * 2048 >> 4)) >> 16 = 4 S, (4 S + 2) >> 2 = S
    assert np.array_equal(cv_resize.resize_linear_u8(im, 1.0), im)
    # constant images stay constant at any scale (coefficients of an axis sum to 2048)
    for v in (0, 1, 127, 255):
        c = np.full((11, 13, 3), v, np.uint8)
        for s in (0.37, 1.9, 3.0, 0.5):
            assert (cv_resize.resize_linear_u8(c, s) == v).all(), (v, s)
    # exact 2 x 2 decimation = INTER_AREA (fast): rounded block mean, ties up ((sum + 2) >> 2), NOT the fixed-point bilinear
    half = cv_resize.resize_linear_u8(im, 0.5)
    blk = im.astype(int).reshape(12, 2, 18, 2, 3).sum(axis=(1, 3))
    assert np.array_equal(half, ((blk + 2) >> 2).astype(np.uint8))
    # odd source sizes: dsize rounds half to even; the ragged last row / column is the float mean of the pixels that exist
    odd = rs.randint(0, 256, (7, 7, 3)).astype(np.uint8)          # 3.5 -> 4: column 3 averages source column 6 alone
    r = cv_resize.resize_linear_u8(odd, 0.5)
    assert r.shape == (4, 4, 3)
    assert np.array_equal(r[3, 3], odd[6, 6])
    assert np.array_equal(r[0, 3], np.rint((odd[0, 6].astype(np.float32) + odd[1, 6]) / np.float32(2)).astype(np.uint8))
    odd5 = rs.randint(0, 256, (5, 5, 3)).astype(np.uint8)         # 2.5 -> 2: the last source row / column is dropped
    assert cv_resize.resize_linear_u8(odd5, 0.5).shape == (2, 2, 3)
    # upscaling by 2: interior samples at quarter positions -> coefficients (1536, 512) / (512, 1536)
    g = np.zeros((1, 4, 3), np.uint8)
    g[0, :, :] = np.array([0, 100, 200, 40])[:, None]
    up = cv_resize.resize_linear_u8(g, 2.0)[0, :, 0]
    assert up.tolist() == [0, 25, 75, 125, 175, 160, 80, 40]


def test_im_prepare_statement_matches_the_reference_worker_lines():
    """im_prepare = flip, crop, resize, BGR->RGB minus means in float64 narrowed to float32, zero padding (data_workers.py:80-121)"""
    rs = np.random.RandomState(2)
    im = rs.randint(0, 256, (20, 30, 3)).astype(np.uint8)
    means = np.array([103.06, 115.90, 123.15])
    rim, (rh, rw) = cv_resize.im_prepare(im, (3, 2, 25, 18), 1.7, True, means, (40, 40))
    res = cv_resize.resize_linear_u8(im[:, ::-1][2:18, 3:25], 1.7)
    assert (rh, rw) == res.shape[:2] == (27, 37)
    for j in range(3):
        want = np.zeros((40, 40), np.float32)
        want[:27, :37] = res[:, :, 2 - j] - means[2 - j]
        assert np.array_equal(rim[j], want)
