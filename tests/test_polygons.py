"""Results.masks.xy (vti_amd/polygons.py): outer-border following + CHAIN_APPROX_SIMPLE restated without OpenCV, checked on
shapes whose cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) answer is known in closed form."""
import numpy as np

from vti_amd.polygons import find_external_contours, masks2segments, scale_coords


def test_rectangle_is_its_four_corners_counter_clockwise_from_top_left():
    m = np.zeros((12, 15), np.uint8)
    m[3:8, 4:11] = 1
    (c,) = find_external_contours(m)
    assert c.tolist() == [[4, 3], [4, 7], [10, 7], [10, 3]]            # (x, y): down the left side first, as OpenCV


def test_single_pixel_line_and_diagonal():
    m = np.zeros((6, 6), np.uint8)
    m[2, 3] = 1
    assert find_external_contours(m)[0].tolist() == [[3, 2]]
    m[:] = 0
    m[1, 1:5] = 1                                                      # a 1-pixel-high bar: out and back, two end points
    assert find_external_contours(m)[0].tolist() == [[1, 1], [4, 1]]
    m[:] = 0
    for i in range(4):
        m[1 + i, 1 + i] = 1                                            # an 8-connected diagonal is ONE component
    cs = find_external_contours(m)
    assert len(cs) == 1 and cs[0].tolist() == [[1, 1], [4, 4]]


def test_holes_are_ignored_and_components_come_in_raster_order():
    m = np.zeros((20, 30), np.uint8)
    m[2:12, 2:12] = 1
    m[5:8, 5:8] = 0                                                    # a hole: RETR_EXTERNAL does not report it
    m[14:17, 20:28] = 1
    cs = find_external_contours(m)
    assert [c.tolist() for c in cs] == [[[2, 2], [2, 11], [11, 11], [11, 2]], [[20, 14], [20, 16], [27, 16], [27, 14]]]


def test_l_shape_and_largest_strategy():
    m = np.zeros((2, 16, 16), np.uint8)
    m[0, 2:10, 2:5] = 1
    m[0, 7:10, 2:12] = 1                                               # an L: the 8-connected border cuts the concave corner diagonally
    m[0, 13, 13] = 1                                                   # + a stray pixel: "largest" keeps the L
    segs = masks2segments(m)
    assert segs[0].dtype == np.float32
    assert segs[0].tolist() == [[2, 2], [2, 9], [11, 9], [11, 7], [5, 7], [4, 6], [4, 2]]      # (5,7) -> (4,6): one diagonal chain step, as cv2
    assert segs[1].shape == (0, 2)                                     # empty mask -> empty polygon
    both = masks2segments(m[:1], strategy="concat")[0]
    assert len(both) == 8


def test_scale_coords_undoes_the_letterbox():
    # 1280x960 frame at imgsz 960 -> 736x960 letterbox (gain 0.75, pad 0 / 8 rows)
    pts = np.array([[0, 8], [960, 728], [480, 368]], np.float32)
    out = scale_coords((736, 960), pts, (960, 1280))
    assert np.allclose(out, [[0, 0], [1280, 960], [640, 480]])
    assert scale_coords((736, 960), np.zeros((0, 2), np.float32), (960, 1280)).shape == (0, 2)


def test_traced_border_is_exactly_the_mask_boundary_on_random_blobs():
    """Every vertex is a foreground pixel with a background 4- or 8-neighbour, and filling between the vertices' extremes
    covers the component's bounding box: a sanity net for shapes without a closed form."""
    rng = np.random.default_rng(0)
    for _ in range(20):
        m = np.zeros((40, 40), np.uint8)
        y, x = rng.integers(5, 25, 2)
        m[y:y + rng.integers(2, 12), x:x + rng.integers(2, 12)] = 1
        m[y + 1:y + 4, x - 3:x + 15] |= 1
        (c,) = find_external_contours(m)
        ys, xs = np.nonzero(m)
        assert c[:, 0].min() == xs.min() and c[:, 0].max() == xs.max() and c[:, 1].min() == ys.min() and c[:, 1].max() == ys.max()
        pad = np.pad(m, 1)
        for px, py in c:
            assert m[py, px] == 1 and pad[py:py + 3, px:px + 3].sum() < 9
