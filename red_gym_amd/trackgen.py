"""Random closed tracks for domain randomisation.

`create_centerline` produces the centre line of the reference's generator
(gym/f110_gym/unittest/random_trackgen.py:56-159, itself adapted from CarRacing-v0): 16 random gates on a ring, a
point that steers towards the next gate in strides of 3.5 units, the last closed lap cut out.  It consumes NumPy's
legacy `RandomState` stream in the reference's draw order and performs the same floating-point operations, so a
seed gives the centre line that `np.random.seed(seed)` gives there.

The reference then offsets the centre line by +-10 units with shapely, strokes both curves with matplotlib
(linewidth 3, 20x20 in at 80 dpi) and re-reads the PNG with cv2 (:161-218).  None of that is reproducible here
(cv2 / shapely absent, matplotlib's anti-aliased rendering is not a specification), so the walls are drawn by
their definition instead: a pixel is a wall iff its distance to the centre line is within half a stroke of the
offset -- `track_mask_kernel` on the GPU, followed by the device map pipeline.  PARITY UNPINNED for the image;
the centre line follows the reference's arithmetic operation by operation.
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


def _checkpoints(rng):
    """Sixteen gates on a ring (:63-77): gate c sits at angle 2*pi*c/16 plus a random part of one sector, at a random
    radius between a third of and the full track radius; the first and the last gate are pinned far out on the
    positive x side so that the lap closes there.  The 32 uniforms are drawn as one [16, 2] block -- RandomState
    fills it row by row, i.e. (angle, radius) per gate, the reference's draw order, and `lo + (hi - lo) * u` is how
    RandomState.uniform maps them."""
    sector = 2 * math.pi / CHECKPOINTS
    u = rng.random_sample((CHECKPOINTS, 2))
    gates = []
    for c in range(CHECKPOINTS):
        angle = 2 * math.pi * c / CHECKPOINTS + (0.0 + (sector - 0.0) * u[c, 0])
        radius = TRACK_RAD / 3 + (TRACK_RAD - TRACK_RAD / 3) * u[c, 1]
        if c == 0:
            angle, radius = 0, 1.5 * TRACK_RAD
        elif c == CHECKPOINTS - 1:
            angle, radius = 2 * math.pi * c / CHECKPOINTS, 1.5 * TRACK_RAD
        gates.append((angle, radius * math.cos(angle), radius * math.sin(angle)))
    return gates, 2 * math.pi * (-0.5) / CHECKPOINTS


def _walk(gates):
    """The walker of :79-129: a point starts at the first gate heading along +y and moves in strides of
    TRACK_DETAIL_STEP; every stride it turns towards the next gate ahead of its polar angle, by at most
    TRACK_TURN_RATE.  Returns per stride (polar angle used for gate selection, mean heading, x, y), for at most
    2500 strides or five completed laps."""
    n = len(gates)
    x, y, heading = 1.5 * TRACK_RAD, 0, 0
    gate, laps, been_below = 0, 0, False
    trail = []
    for _ in range(2500):
        polar = math.atan2(y, x)
        if been_below and polar > 0:
            laps, been_below = laps + 1, False
        if polar < 0:
            been_below = True
            polar += 2 * math.pi
        # first gate at or beyond the polar angle; when the scan runs off the end of the ring it restarts one turn lower
        target = None
        while target is None:
            while True:
                g = gates[gate % n]
                if polar <= g[0]:
                    target = g
                    break
                gate += 1
                if gate % n == 0:
                    break
            if target is None:
                polar -= 2 * math.pi
        ux, uy = math.cos(heading), math.sin(heading)      # unit vector at `heading`; the point moves along its normal
        along = ux * (target[1] - x) + uy * (target[2] - y)
        while heading - polar > 1.5 * math.pi:
            heading -= 2 * math.pi
        while heading - polar < -1.5 * math.pi:
            heading += 2 * math.pi
        before = heading
        along *= SCALE
        if along > 0.3:
            heading -= min(TRACK_TURN_RATE, abs(0.001 * along))
        if along < -0.3:
            heading += min(TRACK_TURN_RATE, abs(0.001 * along))
        x += -uy * TRACK_DETAIL_STEP
        y += ux * TRACK_DETAIL_STEP
        trail.append((polar, before * 0.5 + heading * 0.5, x, y))
        if laps > 4:
            break
    return trail


def create_centerline(rng):
    """random_trackgen.py:56-159.  rng: np.random.RandomState.  Returns the closed centre line [N, 2] in track units,
    or None where the reference gives up (caller retries)."""
    gates, start_angle = _checkpoints(rng)
    trail = np.asarray(_walk(gates), dtype=np.float64)
    # the last closed lap (:131-146): the two latest strides whose polar angle steps over the start angle
    polar = trail[:, 0]
    crossings = np.flatnonzero((polar[1:] > start_angle) & (polar[:-1] <= start_angle)) + 1
    if len(crossings) < 2:
        return None
    first, last = int(crossings[-2]), int(crossings[-1])
    lap = trail[first:last - 1]
    if len(lap) < 3:
        return None
    # head and tail must meet within one stride, measured across the first stride's heading (:149-158)
    px, py = math.cos(lap[0, 1]), math.sin(lap[0, 1])
    gap = np.sqrt(np.square(px * (lap[0, 2] - lap[-1, 2])) + np.square(py * (lap[0, 3] - lap[-1, 3])))
    if gap > TRACK_DETAIL_STEP:
        return None
    return np.ascontiguousarray(lap[:, 2:4])


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
    with torch.cuda.device(free.device):  # stateless entry point: launches on the current device's stream
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
