"""Host half of the track generator (red_gym_amd/trackgen.py; reference unittest/random_trackgen.py:56-159).
The reference module cannot be imported (cv2 / shapely at import time, argparse and mkdir side effects), so these
are invariants of its algorithm, not vectors from it: parity unpinned."""
import numpy as np

from red_gym_amd import trackgen as tg


def test_centerline_is_closed_deterministic_and_evenly_stepped():
    for seed in (123, 1, 2, 3, 7, 99):
        c = tg.random_centerline(seed)
        assert np.array_equal(c, tg.random_centerline(seed))
        assert c.ndim == 2 and c.shape[1] == 2 and 150 < len(c) < 600
        steps = np.linalg.norm(np.diff(c, axis=0), axis=1)
        assert np.allclose(steps, tg.TRACK_DETAIL_STEP, atol=1e-9)          # :125-126: fixed stride
        gap = np.linalg.norm(c[0] - c[-1])
        assert gap <= 2 * tg.TRACK_DETAIL_STEP + 1e-9                        # head and tail glued (:149-158)
        # one counter-clockwise lap around the origin (checkpoints ascend in angle, :66-77)
        ang = np.unwrap(np.arctan2(c[:, 1], c[:, 0]))
        assert 1.8 * np.pi < ang[-1] - ang[0] < 2.2 * np.pi
        r = np.linalg.norm(c, axis=1)
        assert r.min() > tg.TRACK_RAD / 3 - 25 and r.max() < 1.5 * tg.TRACK_RAD + 25
    assert not np.array_equal(tg.random_centerline(1), tg.random_centerline(2))


def test_rng_draw_order_matches_the_reference_statement():
    """Two uniforms per checkpoint, angle first (:67-68): the walker consumes exactly 32 draws per attempt."""
    class Counting(np.random.RandomState):
        n = 0

        def uniform(self, *a, **k):
            Counting.n += 1
            return super().uniform(*a, **k)
    rng = Counting(123)
    tg.create_centerline(rng)
    assert Counting.n == 2 * tg.CHECKPOINTS


def test_raster_frame_holds_every_track():
    for seed in range(20):
        c = tg.random_centerline(seed)
        x0, y0 = tg.raster_frame(c)
        span = tg.MAP_PIXELS * tg.UNITS_PER_PIXEL
        assert (c[:, 0] - tg.WIDTH - 2 > x0).all() and (c[:, 0] + tg.WIDTH + 2 < x0 + span).all()
        assert (c[:, 1] - tg.WIDTH - 2 > y0).all() and (c[:, 1] + tg.WIDTH + 2 < y0 + span).all()
