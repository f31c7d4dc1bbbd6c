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
    """The reference draws two uniforms per gate, angle first (:67-68), with RandomState.uniform; the generator
    takes them as one [16, 2] block of random_sample.  Same stream, same values, same generator state after."""
    import math
    for seed in (0, 123, 4242):
        a, b = np.random.RandomState(seed), np.random.RandomState(seed)
        gates, start_angle = tg._checkpoints(a)
        for c in range(tg.CHECKPOINTS):
            ang = 2 * math.pi * c / tg.CHECKPOINTS + b.uniform(0, 2 * math.pi * 1 / tg.CHECKPOINTS)
            rad = b.uniform(tg.TRACK_RAD / 3, tg.TRACK_RAD)
            if c == 0:
                ang, rad = 0, 1.5 * tg.TRACK_RAD
            if c == tg.CHECKPOINTS - 1:
                ang, rad = 2 * math.pi * c / tg.CHECKPOINTS, 1.5 * tg.TRACK_RAD
            assert gates[c] == (ang, rad * math.cos(ang), rad * math.sin(ang))
        assert start_angle == 2 * math.pi * (-0.5) / tg.CHECKPOINTS
        assert a.random_sample() == b.random_sample()


def test_raster_frame_holds_every_track():
    for seed in range(20):
        c = tg.random_centerline(seed)
        x0, y0 = tg.raster_frame(c)
        span = tg.MAP_PIXELS * tg.UNITS_PER_PIXEL
        assert (c[:, 0] - tg.WIDTH - 2 > x0).all() and (c[:, 0] + tg.WIDTH + 2 < x0 + span).all()
        assert (c[:, 1] - tg.WIDTH - 2 > y0).all() and (c[:, 1] + tg.WIDTH + 2 < y0 + span).all()
