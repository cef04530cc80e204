"""CPU: FocusChip generation (sniper_amd/chips_inference.py, restating lib/chips/chips_inference.py without OpenCV) on
hand-derived cases -- parity unpinned (no cv2 here to mint vectors), so these pin OUR documented semantics."""
import numpy as np

from sniper_amd.chips_inference import _bounding_rects, _dilate, gmask


def test_dilate_and_contour_rects():
    a = np.zeros((7, 7), np.uint8)
    a[3, 3] = 1
    assert _dilate(a, 3).sum() == 9 and _dilate(a, 3)[2:5, 2:5].all()
    assert sorted(zip(*_dilate(a, 2).nonzero())) == [(3, 3), (3, 4), (4, 3), (4, 4)]     # cv2 anchors a 2x2 kernel at (1, 1)
    ring = np.zeros((9, 9), np.uint8)
    ring[2:7, 2:7] = 255
    ring[4, 4] = 0
    # RETR_LIST: the component's outer border and the hole's border (the foreground ring around the hole)
    assert sorted(_bounding_rects(ring)) == [(2, 2, 5, 5), (3, 3, 3, 3)]
    two = np.zeros((6, 10), np.uint8)
    two[1, 1] = two[2, 2] = 255            # diagonal neighbours are one 8-connected component
    two[4, 8] = 255
    assert sorted(_bounding_rects(two)) == [(1, 1, 2, 2), (8, 4, 1, 1)]


def test_gmask_chips():
    m = np.zeros((30, 40), np.float32)            # a 480 x 640 crop at scale 1: 30 x 40 cells of 16 px
    m[5:8, 6:9] = 0.9
    m[20:25, 30:38] = 0.6
    chips = gmask(m, 3, 0.5, ms=4, im_width=640, im_height=480, cscale=1.0)
    # blob 1: cells x 5..9, y 4..8 after 3x3 dilation (5x5) -> >= 4 cells already; blob 2: x 29..38, y 19..25
    assert chips == [[80.0, 64.0, 160.0, 144.0], [464.0, 304.0, 624.0, 416.0]]
    # minimum chip side: a single hot cell grows to ms cells, clamped to the map, scaled back by cscale
    m = np.zeros((30, 40), np.float32)
    m[0, 39] = 1.0
    chips = gmask(m, 1, 0.5, ms=16, im_width=640, im_height=480, cscale=2.0)
    assert chips == [[(640 - 256) / 2.0, 0.0, 320.0, 128.0]]
    # nothing above the threshold: no chips
    assert gmask(np.zeros((10, 10), np.float32), 3, 0.5, ms=4, im_width=160, im_height=160) == []
    # two blobs whose minimum-size chips overlap merge into one chip on the second pass
    m = np.zeros((30, 40), np.float32)
    m[10, 10] = m[10, 14] = 1.0
    chips = gmask(m, 1, 0.5, ms=8, im_width=640, im_height=480, cscale=1.0)
    assert len(chips) == 1 and chips[0][0] <= 10 * 16 - 3 * 16 and chips[0][2] >= 15 * 16
