"""FocusChip generation pinned to OpenCV's published algorithms (row a17): oracle/cv_contours.py restates cv2.dilate, cv2.findContours
(RETR_LIST; Suzuki & Abe's border following, OpenCV contours.cpp) and cv2.boundingRect; the product (csrc/host_inference.cpp) follows
borders too (cv2's rectangles in cv2's order) and keeps the connected-component form of rounds 2-4 as a second, independent route to the
same rectangles.  Host code: runs without a GPU.  (No cv2 in this image: "pinned to the published algorithm".)"""
import numpy as np

from oracle import cv_contours as cvc


def _rects(mask, mode):
    from sniper_amd import hip
    out, n = np.zeros((4096, 4), np.int32), np.zeros(1, np.int32)
    m = np.ascontiguousarray(mask, np.uint8)
    hip.call('sn_focus_rects_host', m, m.shape[0], m.shape[1], mode, out, 4096, n)
    return [tuple(int(v) for v in r) for r in out[:int(n[0])]]


def _random_mask(rs, it):
    h, w = int(rs.randint(1, 30)), int(rs.randint(1, 40))
    if it % 2:
        return (rs.rand(h, w) < rs.choice([0.05, 0.2, 0.5, 0.8])).astype(np.uint8) * 255
    m = np.zeros((h, w), np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(rs.randint(1, 6)):
        cy, cx, r = rs.randint(0, h), rs.randint(0, w), rs.randint(1, max(2, min(h, w) // 3))
        m[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 255
    for _ in range(rs.randint(0, 4)):                       # punch holes (some open to the outside, some enclosed)
        cy, cx = rs.randint(0, h), rs.randint(0, w)
        m[max(cy - 1, 0):cy + rs.randint(1, 3), max(cx - 1, 0):cx + rs.randint(1, 3)] = 0
    return m


def test_hand_cases_of_the_border_following():
    ring = np.zeros((9, 9), np.uint8)
    ring[2:7, 2:7] = 255
    ring[4, 4] = 0
    cs = cvc.find_contours_list(ring)
    # RETR_LIST: the outer border of the square (16 border cells) and the hole border (the 4 cells 4-adjacent to the hole, traced as
    # an 8-connected cycle); newest first: the hole border, found on row 4, before the outer border found on row 2
    assert [cvc.bounding_rect(c) for c in cs] == [(3, 3, 3, 3), (2, 2, 5, 5)]
    assert len(cs[1]) == 16 and sorted(map(tuple, cs[0])) == [(3, 4), (4, 3), (4, 5), (5, 4)]
    dot = np.zeros((3, 4), np.uint8)
    dot[1, 2] = 7
    assert [c.tolist() for c in cvc.find_contours_list(dot)] == [[[2, 1]]]
    two = np.zeros((6, 10), np.uint8)
    two[1, 1] = two[2, 2] = 255                              # diagonal neighbours: one 8-connected component
    two[4, 8] = 255
    assert [cvc.bounding_rect(c) for c in cvc.find_contours_list(two)] == [(8, 4, 1, 1), (1, 1, 2, 2)]
    full = np.full((4, 5), 255, np.uint8)                    # foreground touching every image border: one outer border
    assert [cvc.bounding_rect(c) for c in cvc.find_contours_list(full)] == [(0, 0, 5, 4)]
    assert cvc.find_contours_list(np.zeros((3, 3), np.uint8)) == []


def test_dilate_restatement():
    a = np.zeros((7, 7), np.uint8)
    a[3, 3] = 1
    assert cvc.dilate_rect(a, 3).sum() == 9 and cvc.dilate_rect(a, 3)[2:5, 2:5].all()
    assert sorted(zip(*cvc.dilate_rect(a, 2).nonzero())) == [(3, 3), (3, 4), (4, 3), (4, 4)]      # anchor (1, 1) of a 2 x 2 kernel
    from sniper_amd.chips_inference import _dilate
    rs = np.random.RandomState(3)
    for _ in range(300):
        h, w, d = rs.randint(1, 20), rs.randint(1, 20), rs.randint(1, 6)
        m = (rs.rand(h, w) < 0.15).astype(np.uint8)
        assert np.array_equal(cvc.dilate_rect(m, d), _dilate(m, d))


def test_border_following_components_and_native_code_agree_on_random_masks():
    from sniper_amd.chips_inference import _bounding_rects
    rs = np.random.RandomState(1)
    holes = 0
    for it in range(1500):
        m = _random_mask(rs, it)
        want = [cvc.bounding_rect(c) for c in cvc.find_contours_list(m)]
        native_borders, native_components = _rects(m, 0), _rects(m, 1)
        statement = [tuple(int(v) for v in r) for r in _bounding_rects(m)]
        assert native_borders == want                        # cv2's rectangles in cv2's order
        assert native_components == statement                # the component form, scipy's order
        assert sorted(native_borders) == sorted(native_components)
        holes += len(want) > len(set(want)) or any(r[2] >= 3 and r[3] >= 3 for r in want)
    assert holes > 100


def test_gmask_native_equals_the_restated_reference_lines():
    """sn_focus_chips_host (what Tester uses) == chips_inference.py:12-89 over the restated cv2 calls, chip order included; and, as a set
    of chips, == the scipy.ndimage statement kept in sniper_amd/chips_inference.py."""
    from sniper_amd.chips_inference import gmask, gmask_reference
    rs = np.random.RandomState(2)
    multi = 0
    for it in range(300):
        h, w = int(rs.randint(8, 60)), int(rs.randint(8, 80))
        mp = np.zeros((h, w), np.float32)
        yy, xx = np.mgrid[0:h, 0:w]
        for _ in range(rs.randint(1, 6)):
            cy, cx, r = rs.randint(0, h), rs.randint(0, w), rs.randint(1, 6)
            mp[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = rs.uniform(0.3, 1.0)
        d, ms, cs = int(rs.choice([1, 2, 3, 5])), int(rs.choice([2, 4, 8, 16])), float(rs.choice([1.0, 1.6667, 2.9167]))
        iw, ih = w * 16 - int(rs.randint(0, 16)), h * 16 - int(rs.randint(0, 16))
        a, b, c = gmask(mp, d, 0.5, ms, iw, ih, cs), cvc.gmask(mp, d, 0.5, ms, iw, ih, cs), gmask_reference(mp, d, 0.5, ms, iw, ih, cs)
        assert a == b, (it, a, b)
        assert sorted(map(tuple, a)) == sorted(map(tuple, c))
        multi += len(a) > 1
    assert multi > 50
