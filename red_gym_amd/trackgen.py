"""Random closed tracks for domain randomisation.

`create_centerline` restates the track walker of the reference's generator
(gym/f110_gym/unittest/random_trackgen.py:56-159, itself adapted from CarRacing-v0): 16 random checkpoints on a
ring, a point that steers towards the next checkpoint in steps of 3.5 units, the last closed lap cut out.  It
draws from NumPy's legacy global-style `RandomState` in the reference's call order, so a seed gives the same
centre line as `np.random.seed(seed)` does there.

The reference then offsets the centre line by +-10 units with shapely, strokes both curves with matplotlib
(linewidth 3, 20x20 in at 80 dpi) and re-reads the PNG with cv2 (:161-218).  None of that is reproducible here
(cv2 / shapely absent, matplotlib's anti-aliased rendering is not a specification), so the walls are drawn by
their definition instead: a pixel is a wall iff its distance to the centre line is within half a stroke of the
offset -- `track_mask_kernel` on the GPU, followed by the device map pipeline.  PARITY UNPINNED for the image;
the centre line follows the reference's arithmetic statement by statement.
"""
import math

import numpy as np

CHECKPOINTS = 16
SCALE = 6.0
TRACK_RAD = 900 / SCALE
TRACK_DETAIL_STEP = 21 / SCALE
TRACK_TURN_RATE = 0.31
WIDTH = 10.0                      # random_trackgen.py:55: offset of the walls from the centre line, in track units

# the reference's figure: 1600x1600 px for x in (-180, 300), y in (-300, 300), aspect equal (:185-193), yaml
# resolution 0.0625 m/px (:222): the y extent fills the height, so one track unit is 1600/600 px
MAP_PIXELS = 1600
UNITS_PER_PIXEL = 600.0 / MAP_PIXELS
RESOLUTION = 0.0625
METRES_PER_UNIT = RESOLUTION / UNITS_PER_PIXEL
STROKE_PIXELS = 3 * 80 / 72.0     # linewidth 3 pt at 80 dpi


def create_centerline(rng):
    """random_trackgen.py:56-159.  rng: np.random.RandomState (or the np.random module).  Returns the closed centre
    line [N, 2] in track units, or None where the reference returns False (caller retries)."""
    start_alpha = 0.
    checkpoints = []
    for c in range(CHECKPOINTS):
        alpha = 2 * math.pi * c / CHECKPOINTS + rng.uniform(0, 2 * math.pi * 1 / CHECKPOINTS)
        rad = rng.uniform(TRACK_RAD / 3, TRACK_RAD)
        if c == 0:
            alpha = 0
            rad = 1.5 * TRACK_RAD
        if c == CHECKPOINTS - 1:
            alpha = 2 * math.pi * c / CHECKPOINTS
            start_alpha = 2 * math.pi * (-0.5) / CHECKPOINTS
            rad = 1.5 * TRACK_RAD
        checkpoints.append((alpha, rad * math.cos(alpha), rad * math.sin(alpha)))

    x, y, beta = 1.5 * TRACK_RAD, 0, 0
    dest_i = 0
    laps = 0
    track = []
    no_freeze = 2500
    visited_other_side = False
    while True:
        alpha = math.atan2(y, x)
        if visited_other_side and alpha > 0:
            laps += 1
            visited_other_side = False
        if alpha < 0:
            visited_other_side = True
            alpha += 2 * math.pi
        while True:
            failed = True
            while True:
                dest_alpha, dest_x, dest_y = checkpoints[dest_i % len(checkpoints)]
                if alpha <= dest_alpha:
                    failed = False
                    break
                dest_i += 1
                if dest_i % len(checkpoints) == 0:
                    break
            if not failed:
                break
            alpha -= 2 * math.pi
            continue
        r1x = math.cos(beta)
        r1y = math.sin(beta)
        p1x = -r1y
        p1y = r1x
        dest_dx = dest_x - x
        dest_dy = dest_y - y
        proj = r1x * dest_dx + r1y * dest_dy
        while beta - alpha > 1.5 * math.pi:
            beta -= 2 * math.pi
        while beta - alpha < -1.5 * math.pi:
            beta += 2 * math.pi
        prev_beta = beta
        proj *= SCALE
        if proj > 0.3:
            beta -= min(TRACK_TURN_RATE, abs(0.001 * proj))
        if proj < -0.3:
            beta += min(TRACK_TURN_RATE, abs(0.001 * proj))
        x += p1x * TRACK_DETAIL_STEP
        y += p1y * TRACK_DETAIL_STEP
        track.append((alpha, prev_beta * 0.5 + beta * 0.5, x, y))
        if laps > 4:
            break
        no_freeze -= 1
        if no_freeze == 0:
            break

    # the last closed lap (:131-146)
    i1, i2 = -1, -1
    i = len(track)
    while True:
        i -= 1
        if i == 0:
            return None
        pass_through_start = track[i][0] > start_alpha and track[i - 1][0] <= start_alpha
        if pass_through_start and i2 == -1:
            i2 = i
        elif pass_through_start and i1 == -1:
            i1 = i
            break
    track = track[i1:i2 - 1]
    if len(track) < 3:
        return None
    first_beta = track[0][1]
    first_perp_x = math.cos(first_beta)
    first_perp_y = math.sin(first_beta)
    well_glued_together = np.sqrt(np.square(first_perp_x * (track[0][2] - track[-1][2])) +
                                  np.square(first_perp_y * (track[0][3] - track[-1][3])))
    if well_glued_together > TRACK_DETAIL_STEP:
        return None
    return np.asarray([(px, py) for (_, _, px, py) in track], dtype=np.float64)


def random_centerline(seed, max_tries=64):
    """The reference's retry loop (:228-234): first centre line that closes, from RandomState(seed)."""
    rng = np.random.RandomState(seed)
    for _ in range(max_tries):
        c = create_centerline(rng)
        if c is not None:
            return c
    raise RuntimeError('no closed track in %d tries (seed %d)' % (max_tries, seed))


class Track(object):
    """A generated track in world metres: occupancy mask on the device, centre-line waypoints, map origin."""
    __slots__ = ('free', 'resolution', 'orig_x', 'orig_y', 'waypoints', 'centerline_units')


def raster_frame(centerline_units):
    """World frame of the reference's figure (:185-205): pixel (0,0) is the lower-left corner of the 1600x1600
    canvas spanning x in (-180, 300) horizontally centred, y in (-300, 300); the first centre-line point is the
    world origin.  Returns (x0_units, y0_units) of that corner."""
    span_x = MAP_PIXELS * UNITS_PER_PIXEL
    x0 = (-180 + 300) / 2.0 - span_x / 2.0
    y0 = -300.0
    return x0, y0


def generate(seed, device='cuda', width_units=WIDTH, stroke_pixels=STROKE_PIXELS):
    """Random track -> Track with the mask drawn on `device`.  Walls: pixels whose centre lies within half a
    stroke of the curves at distance `width_units` from the centre line (both sides)."""
    import ctypes as C
    import torch
    from . import _lib
    lib = _lib.load()
    cl = random_centerline(seed)
    x0, y0 = raster_frame(cl)
    dev = torch.device(device)
    pts = torch.as_tensor(np.ascontiguousarray(cl), device=dev)
    free = torch.empty((MAP_PIXELS, MAP_PIXELS), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(lib.f110_track_mask(pts.data_ptr(), len(cl), 1, MAP_PIXELS, MAP_PIXELS, x0, y0, UNITS_PER_PIXEL,
                                   float(width_units), 0.5 * stroke_pixels * UNITS_PER_PIXEL, free.data_ptr(), stream))
    t = Track()
    t.free = free
    t.resolution = RESOLUTION
    # world metres: origin at the first centre-line point (:201-205)
    t.orig_x = (x0 - cl[0, 0]) * METRES_PER_UNIT
    t.orig_y = (y0 - cl[0, 1]) * METRES_PER_UNIT
    wp = (cl - cl[0]) * METRES_PER_UNIT
    heading = np.arctan2(np.roll(wp[:, 1], -1) - wp[:, 1], np.roll(wp[:, 0], -1) - wp[:, 0])
    t.waypoints = np.column_stack([wp, heading])
    t.centerline_units = cl
    return t
