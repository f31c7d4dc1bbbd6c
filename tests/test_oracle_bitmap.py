"""CPU checks of oracle/lidar_bitmap.c (the restatement of OpenCV 4.11's drawing code that the bitmap kernel is
held to).  cv2 is not importable here, so these are hand-derived known answers, invariants of the published
algorithms, and a coarse cross-check against Pillow's independent polygon filler -- not a pin (parity unpinned)."""
import numpy as np
import pytest

from oracle import bitmap as ob


def _px(img):
    return sorted(map(tuple, np.argwhere(img.T > 0).tolist()))   # (x, y) pairs


def test_bresenham_known_answers():
    # err = dx - 2dy; minor step when err < 0 (LineIterator, connectivity 8)
    assert _px(ob.line((8, 8), (0, 0), (5, 2))) == [(0, 0), (1, 0), (2, 1), (3, 1), (4, 2), (5, 2)]
    assert _px(ob.line((8, 8), (1, 1), (1, 6))) == [(1, y) for y in range(1, 7)]
    assert _px(ob.line((8, 8), (6, 3), (0, 3))) == [(x, 3) for x in range(7)]
    assert _px(ob.line((8, 8), (0, 0), (7, 7))) == [(i, i) for i in range(8)]
    assert _px(ob.line((8, 8), (2, 7), (4, 0))) == sorted([(2, 7), (2, 6), (3, 5), (3, 4), (3, 3), (3, 2), (4, 1), (4, 0)])
    assert _px(ob.line((4, 4), (2, 2), (2, 2))) == [(2, 2)]


def test_line_is_direction_independent_and_clipped():
    rng = np.random.default_rng(1)
    for _ in range(300):
        a, b = rng.integers(-40, 80, 2), rng.integers(-40, 80, 2)
        f, g = ob.line((40, 48), a, b), ob.line((40, 48), b, a)
        inside = (0 <= a[0] < 48 and 0 <= a[1] < 40, 0 <= b[0] < 48 and 0 <= b[1] < 40)
        if a[0] != b[0] and all(inside):     # leftToRight makes the walk start at the left end (clipping is not symmetric)
            assert np.array_equal(f, g)
        if inside[0]:
            assert f[a[1], a[0]]
        if inside[1]:
            assert f[b[1], b[0]]
        assert f.sum() // 255 <= max(abs(a[0] - b[0]), abs(a[1] - b[1])) + 1


def test_rectangle_is_the_inclusive_box():
    img = ob.rectangle_filled((10, 12), (3, 2), (7, 6))
    want = np.zeros((10, 12), np.uint8); want[2:7, 3:8] = 255
    assert np.array_equal(img, want)
    img = ob.rectangle_filled((10, 12), (-2, -2), (2, 2))
    want = np.zeros((10, 12), np.uint8); want[0:3, 0:3] = 255
    assert np.array_equal(img, want)
    assert ob.rectangle_filled((10, 12), (20, 20), (24, 24)).sum() == 0


def test_fill_poly_simple_shapes():
    sq = ob.fill_poly((10, 10), [(2, 2), (7, 2), (7, 6), (2, 6)])
    want = np.zeros((10, 10), np.uint8); want[2:7, 2:8] = 255
    assert np.array_equal(sq, want)
    # a polygon that surrounds the image fills it; one that misses it leaves it alone
    assert (ob.fill_poly((16, 16), [(-50, -50), (70, -50), (70, 70), (-50, 70)]) == 255).all()
    assert ob.fill_poly((16, 16), [(30, 30), (60, 30), (60, 60)]).sum() == 0
    # orientation does not matter (even-odd rule)
    tri = [(1, 1), (10, 3), (4, 10)]
    assert np.array_equal(ob.fill_poly((12, 12), tri), ob.fill_poly((12, 12), tri[::-1]))
    # the fill contains the outline drawn by polylines
    assert ((ob.polylines((12, 12), tri) > 0) <= (ob.fill_poly((12, 12), tri) > 0)).all()


def test_fill_poly_close_to_pillow():
    from PIL import Image, ImageDraw
    rng = np.random.default_rng(2)
    th = np.linspace(0, 2 * np.pi, 1080)
    for trial in range(6):
        scan = np.clip(6 + 4 * np.sin(3 * th + trial) + rng.normal(0, 0.2, 1080), 0, 30) * (1 + (trial == 5))
        pts = ob.points(scan)
        mine = ob.fill_poly((256, 256), pts) > 0
        pil = Image.new('L', (256, 256), 0)
        ImageDraw.Draw(pil).polygon([tuple(p) for p in pts.tolist()], fill=255, outline=255)
        pil = np.array(pil) > 0
        outline = ob.polylines((256, 256), pts) > 0
        near = outline.copy()                 # 1-px dilation of the outline
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                near |= np.roll(np.roll(outline, dy, 0), dx, 1)
        assert not ((mine != pil) & ~near).any()      # the two fillers differ only on boundary pixels
        assert abs(int(mine.sum()) - int(pil.sum())) < 0.02 * mine.sum() + 50


def test_lidar_to_bitmap_reference_semantics():
    scan = 3 + np.sin(np.linspace(0, 9, 1080)) ** 2
    idx, cs, sn = ob.tables(1080, 600)
    assert idx[0] == 0 and idx[-1] == 1079 and (np.diff(idx) >= 1).all() and len(idx) == 600
    pts = ob.points(scan)
    k = 123
    assert pts[k, 0] == int(np.rint(128 + 10 * scan[idx[k]] * cs[k])) and pts[k, 1] == int(np.rint(128 + 10 * scan[idx[k]] * sn[k]))
    g = ob.lidar_to_bitmap(scan, bg_color='black', draw_mode='FILL')
    assert g.shape == (256, 256) and set(np.unique(g)) == {0, 255}
    assert g[126:131, 126:131].max() == 0                         # centre marker in background colour (lidar.py:99)
    p = ob.lidar_to_bitmap(scan, bg_color='black', draw_mode='POLYGON')
    assert p[126:131, 126:131].min() == 255                       # ... in draw colour otherwise
    w = ob.lidar_to_bitmap(scan, bg_color='white', draw_mode='FILL')
    assert np.array_equal(w, 255 - g)
    c3 = ob.lidar_to_bitmap(scan, bg_color='black', draw_mode='FILL', channels=3)
    c4 = ob.lidar_to_bitmap(scan, bg_color='black', draw_mode='FILL', channels=4)
    assert c3.shape == (256, 256, 3) and (c3 == g[..., None]).all()
    assert c4.shape == (256, 256, 4) and (c4[..., :3] == g[..., None]).all() and (c4[..., 3] == 255).all()
    cw = ob.lidar_to_bitmap(scan, winding_dir='CW', bg_color='black', draw_mode='FILL')
    assert not np.array_equal(cw, g)
    with pytest.raises(ValueError):
        ob.lidar_to_bitmap(scan, scaling_factor=None)
    assert ob.lidar_to_bitmap(scan, max_scan_radius=25.6, scaling_factor=None, bg_color='black', draw_mode='FILL').tolist() == g.tolist()


def test_occupancy_against_the_python_loop():
    rng = np.random.default_rng(3)
    scan = np.clip(rng.uniform(0, 14, 1080), 0, 30)
    scan[::50] = 30.0
    angles = np.linspace(-135, 135, 1080) * np.pi / 180.0
    want = np.zeros((256, 256), np.uint8)
    for b in range(1080):                                          # f1tenth_gym/examples/lidar.py:222-242
        r = scan[b]
        if r >= 30.0:
            continue
        x, y = r * np.cos(angles[b]), r * np.sin(angles[b])
        if not (-10.0 <= x <= 10.0 and -10.0 <= y <= 10.0):
            continue
        i_row = int(((x + 10.0) / 20.0) * 255)
        i_col = int(((y + 10.0) / 20.0) * 255)
        want[np.clip(i_row, 0, 255), np.clip(i_col, 0, 255)] = 1
    assert np.array_equal(ob.occupancy(scan), want)
