"""GPU half of the track generator: the wall mask against the same expressions in NumPy, and a generated track
driven by the GPU pure-pursuit planner.  (Image parity with the reference's matplotlib/cv2 rendering is unpinned,
red_gym_amd/trackgen.py.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mask_numpy(pts, closed, H, W, x0, y0, pixel, offset, half):
    ix, iy = np.meshgrid(np.arange(W), np.arange(H))
    px, py = x0 + (ix + 0.5) * pixel, y0 + (iy + 0.5) * pixel
    n = len(pts)
    best = np.full((H, W), np.inf)
    for i in range(n if closed else n - 1):
        ax, ay = pts[i]
        ex, ey = pts[(i + 1) % n][0] - ax, pts[(i + 1) % n][1] - ay
        l2 = ex * ex + ey * ey
        inv = 1.0 / l2 if l2 > 0 else 0.0
        rx, ry = px - ax, py - ay
        t = np.clip((rx * ex + ry * ey) * inv, 0.0, 1.0)
        dx, dy = rx - t * ex, ry - t * ey
        best = np.minimum(best, dx * dx + dy * dy)
    return (~(np.abs(np.sqrt(best) - offset) <= half)).astype(np.uint8)


@pytest.mark.parametrize('closed', [1, 0])
def test_track_mask_matches_numpy(closed):
    import ctypes as C
    from red_gym_amd import _lib
    rng = np.random.default_rng(4 + closed)
    pts = np.cumsum(rng.normal(0, 6, (23, 2)), axis=0) + 40
    pts[7] = pts[6]                                            # a zero-length segment
    H, W = 150, 211
    d = torch.as_tensor(pts, device='cuda')
    out = torch.empty((H, W), dtype=torch.uint8, device='cuda')
    _lib.check(_lib.load().f110_track_mask(d.data_ptr(), len(pts), closed, H, W, -10.0, 3.0, 0.45, 4.0, 0.8, out.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream))
    want = _mask_numpy(pts, closed, H, W, -10.0, 3.0, 0.45, 4.0, 0.8)
    got = out.cpu().numpy()
    assert 0.02 < (got == 0).mean() < 0.6
    assert np.array_equal(got, want)


def _track_mask(pts, closed, H, W, x0, y0, pixel, offset, half):
    from red_gym_amd import _lib
    d = torch.as_tensor(np.ascontiguousarray(pts, dtype=np.float64), device='cuda')
    out = torch.empty((H, W), dtype=torch.uint8, device='cuda')
    _lib.check(_lib.load().f110_track_mask(d.data_ptr(), len(pts), closed, H, W, x0, y0, pixel, offset, half, out.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream))
    return out.cpu().numpy()


def _pixel_centres(H, W, x0, y0, pixel):
    ix, iy = np.meshgrid(np.arange(W), np.arange(H))
    return x0 + (ix + 0.5) * pixel, y0 + (iy + 0.5) * pixel


def test_track_walls_of_a_circle_are_two_circles():
    """Analytic case (not the kernel's own formula): a circular centre line of radius R has walls on the circles R - offset
    and R + offset, each half a stroke thick.  Every pixel that is clearly inside a wall ring (by more than a pixel diagonal
    of slack) must be wall, every pixel clearly outside must be free, and both rings must be closed."""
    H = W = 400
    pixel, R, offset, half = 0.375, 45.0, 10.0, 0.625          # the generator's units: 600 / 1600 per pixel, WIDTH 10, 3.33 px stroke
    th = np.linspace(0.0, 2 * np.pi, 720, endpoint=False)
    pts = np.column_stack([75.0 + R * np.cos(th), 75.0 + R * np.sin(th)])
    got = _track_mask(pts, 1, H, W, 0.0, 0.0, pixel, offset, half)
    px, py = _pixel_centres(H, W, 0.0, 0.0, pixel)
    r = np.hypot(px - 75.0, py - 75.0)
    ring = np.minimum(np.abs(r - (R - offset)), np.abs(r - (R + offset)))     # distance to the nearer wall circle
    slack = 0.02                                                # sagitta of the 720-gon (0.0004) + float noise; pixel centres are exact
    assert (got[ring < half - slack] == 0).all()               # inside a wall ring: wall
    assert (got[ring > half + slack] == 1).all()               # outside: free
    # closed rings: walking around each wall circle, every sample point's pixel is wall
    for rad in (R - offset, R + offset):
        a = np.linspace(0, 2 * np.pi, 4000)
        cx = np.floor((75.0 + rad * np.cos(a)) / pixel).astype(int)
        cy = np.floor((75.0 + rad * np.sin(a)) / pixel).astype(int)
        assert (got[cy, cx] == 0).all()
    # and the corridor between them is free along the centre line
    cx = np.floor(pts[:, 0] / pixel).astype(int); cy = np.floor(pts[:, 1] / pixel).astype(int)
    assert (got[cy, cx] == 1).all()


def test_track_walls_of_a_straight_open_line_are_a_stadium():
    """Analytic case: an OPEN straight centre line from A to B: the walls are the two parallels at distance `offset` and the
    two half circles of radius `offset` round the ends (the set at distance `offset` from a segment), half a stroke thick."""
    H, W = 240, 520
    pixel, offset, half = 0.375, 10.0, 0.625
    A, B = np.array([30.0, 45.0]), np.array([160.0, 45.0])
    pts = np.stack([A + (B - A) * t for t in np.linspace(0, 1, 14)])          # collinear points: the same segment
    got = _track_mask(pts, 0, H, W, 0.0, 0.0, pixel, offset, half)
    px, py = _pixel_centres(H, W, 0.0, 0.0, pixel)
    t = np.clip((px - A[0]) / (B[0] - A[0]), 0.0, 1.0)
    dist = np.hypot(px - (A[0] + t * (B[0] - A[0])), py - A[1])               # distance to the segment, closed form
    band = np.abs(dist - offset)
    assert (got[band < half - 1e-6] == 0).all() and (got[band > half + 1e-6] == 1).all()
    # the two parallels
    for yy in (A[1] - offset, A[1] + offset):
        cx = np.floor(np.linspace(A[0], B[0], 900) / pixel).astype(int)
        assert (got[int(np.floor(yy / pixel)), cx] == 0).all()
    # the round caps: points at distance `offset` beyond each end
    for end, sgn in ((A, -1.0), (B, 1.0)):
        a = np.linspace(-np.pi / 2, np.pi / 2, 400)
        cx = np.floor((end[0] + sgn * offset * np.cos(a)) / pixel).astype(int)
        cy = np.floor((end[1] + offset * np.sin(a)) / pixel).astype(int)
        assert (got[cy, cx] == 0).all()


def test_generated_track_walls_keep_their_distance_from_the_centre_line():
    """Invariant on a real generated track, checked with an independent tool: scipy's exact EDT of the rasterised centre
    line.  Every wall pixel lies WIDTH +- (stroke / 2 + 1 px) from the centre line, and every pixel that close to that
    distance band's middle is a wall (so the walls are complete loops on both sides)."""
    from scipy.ndimage import distance_transform_edt
    from red_gym_amd import trackgen
    t = trackgen.generate(11, device='cuda')
    wall = t.free.cpu().numpy() == 0
    cl = t.centerline_units
    x0, y0 = trackgen.raster_frame(cl)
    # rasterise the closed centre line densely (10 samples per pixel of length)
    nxt = np.roll(cl, -1, axis=0)
    seg = np.hypot(*(nxt - cl).T)
    line = np.ones((trackgen.MAP_PIXELS, trackgen.MAP_PIXELS), dtype=bool)
    for a, b, L in zip(cl, nxt, seg):
        n = max(2, int(L / trackgen.UNITS_PER_PIXEL * 10))
        p = a + (b - a) * np.linspace(0, 1, n)[:, None]
        ix = np.floor((p[:, 0] - x0) / trackgen.UNITS_PER_PIXEL).astype(int)
        iy = np.floor((p[:, 1] - y0) / trackgen.UNITS_PER_PIXEL).astype(int)
        ok = (ix >= 0) & (ix < trackgen.MAP_PIXELS) & (iy >= 0) & (iy < trackgen.MAP_PIXELS)
        line[iy[ok], ix[ok]] = False
    d_px = distance_transform_edt(line)                                    # pixels to the nearest centre-line pixel
    d_units = d_px * trackgen.UNITS_PER_PIXEL
    half = 0.5 * trackgen.STROKE_PIXELS * trackgen.UNITS_PER_PIXEL
    tol = half + 1.5 * trackgen.UNITS_PER_PIXEL                           # raster of the line (<= 1 px) + pixel-centre offset
    assert wall.sum() > 5000
    assert (np.abs(d_units[wall] - trackgen.WIDTH) <= tol).all()
    core = np.abs(d_units - trackgen.WIDTH) <= max(half - 1.5 * trackgen.UNITS_PER_PIXEL, 0.0)
    # (where the track bends sharply the inner offset curve self-intersects: a pixel can be WIDTH from one stretch of the
    # centre line and nearer to another -- it is then not at distance WIDTH from the LINE; the EDT is the distance to the line,
    # so `core` already excludes those)
    assert core.sum() > 1000 and wall[core].all()


def test_generated_track_is_a_drivable_loop():
    from red_gym_amd import F110VecEnv, workload, trackgen
    B = 256
    env = F110VecEnv(B, map=workload.EXAMPLE_MAP, num_agents=1, autoreset=False)
    t = env.randomize_track(seed=123)
    free = t.free.cpu().numpy()
    assert free.shape == (1600, 1600) and 0.005 < (free == 0).mean() < 0.05      # two thin wall loops
    # the corridor between the walls contains the centre line: every waypoint's cell is free and >= 1 m from a wall
    dt = env.eng.get_map_dt()
    col = np.floor((t.waypoints[:, 0] - t.orig_x) / t.resolution).astype(int)
    row = np.floor((t.waypoints[:, 1] - t.orig_y) / t.resolution).astype(int)
    assert (dt[row, col] > 1.0).all()
    half_width_m = trackgen.WIDTH * trackgen.METRES_PER_UNIT
    assert np.allclose(dt[row, col], half_width_m, atol=0.25)
    # cars spread along the centre line follow it with the GPU planner without touching the walls
    k = (np.arange(B) * 7) % len(t.waypoints)
    poses = t.waypoints[k][:, None, :].copy()
    obs = env.reset(poses)[0]
    assert not bool(obs['collisions'].any())
    wp = torch.as_tensor(np.column_stack([t.waypoints[:, :2], np.full(len(t.waypoints), 3.0)]), device=env.device)
    crashed = torch.zeros(B, dtype=torch.bool, device=env.device)
    for _ in range(400):
        obs = env.step(env.pure_pursuit(wp, 1.5, 1.0))[0]
        crashed |= obs['collisions'][:, 0] > 0
    assert int(crashed.sum()) == 0
    moved = torch.linalg.norm(env.state[:, 0, :2] - torch.as_tensor(poses[:, 0, :2], device=env.device), dim=1)
    assert float(moved.median()) > 5.0                                           # 4 s at ~3 m/s
    # a second seed replaces the map in place
    t2 = env.randomize_track(seed=7)
    assert not torch.equal(t2.free, t.free)
    env.close()


def test_map_slots_give_env_blocks_their_own_map():
    """Blocks of envs on different maps inside one handle scan exactly like separate single-map envs on those maps --
    maps of all four KINDS (resolution a power of two or not, origin rotated or not): the shard is then scanned block by
    block, each block with the instantiation its own maps allow."""
    from red_gym_amd import F110VecEnv, workload, maps
    B = 96
    multi = F110VecEnv(B, map=workload.EXAMPLE_MAP, num_agents=1)
    names = [None, 'berlin', 'skirk', 'rot']
    specs = []
    for k, name in enumerate(names):
        y = workload.EXAMPLE_MAP + '.yaml' if name in (None, 'rot') else maps.builtin_map_yaml(name)
        m = maps.load_map(y, '.png')
        theta = 0.3 if name == 'rot' else float(np.arctan2(m.orig_s, m.orig_c))
        specs.append((m.free, m.resolution, m.orig_x, m.orig_y, theta))
        multi.eng.set_map_occupancy(*specs[-1], slot=k)
    assign = (np.arange(B) * 4) // B
    multi.eng.assign_maps(assign)
    rng = np.random.default_rng(0)
    poses = np.zeros((B, 1, 3))
    poses[:, 0, :2] = rng.uniform(-3, 3, (B, 2))
    poses[:, 0, 2] = rng.uniform(-3, 3, B)
    acts = workload.action_pool(4, B, 1)
    om = multi.reset(poses)[0]
    singles = [F110VecEnv(B, map=workload.EXAMPLE_MAP, num_agents=1) for n in names]
    for sgl, spec in zip(singles, specs):
        sgl.eng.set_map_occupancy(*spec)
    outs = [s.reset(poses)[0] for s in singles]
    for step in range(4):
        for k in range(4):
            sel = torch.as_tensor(assign == k, device=multi.device)
            assert torch.equal(om['scans'][sel], outs[k]['scans'][sel]), (step, k)
            assert torch.equal(om['collisions'][sel], outs[k]['collisions'][sel])
        om = multi.step(acts[step])[0]
        outs = [s.step(acts[step])[0] for s in singles]
    assert np.array_equal(multi.eng.get_map_dt(slot=1), singles[1].eng.get_map_dt())
    # the maps differ, so the blocks really see different worlds
    assert not torch.equal(outs[0]['scans'], outs[1]['scans'])
    # error behaviour: a slot that holds no map
    with pytest.raises(IndexError):
        multi.eng.assign_maps(np.full(B, 5))
    multi.eng.assign_maps(None)
    assert torch.equal(multi.reset(poses)[0]['scans'], singles[0].reset(poses)[0]['scans'])
    for e in [multi] + singles:
        e.close()


def test_a_map_per_env_neighbouring_cars_on_different_maps():
    """f110_assign_maps without the block rule: envs take their maps in any pattern (here env e on map e % 4, an odd number of
    envs, slots far beyond the first 64) -- the scan then runs one wave per workgroup, every car staging its own map's table --
    and every env still scans, collides and steps exactly like a single-map env on its map."""
    from red_gym_amd import F110VecEnv, workload, maps
    from red_gym_amd import _lib
    B = 97
    multi = F110VecEnv(B, map=workload.EXAMPLE_MAP, num_agents=1)
    names = [None, 'berlin', 'skirk', 'rot']
    slots = [0, 65, 1000, _lib.F110_MAX_MAPS - 1]
    specs = []
    for k, name in zip(slots, names):
        y = workload.EXAMPLE_MAP + '.yaml' if name in (None, 'rot') else maps.builtin_map_yaml(name)
        m = maps.load_map(y, '.png')
        theta = 0.3 if name == 'rot' else float(np.arctan2(m.orig_s, m.orig_c))
        specs.append((m.free, m.resolution, m.orig_x, m.orig_y, theta))
        multi.eng.set_map_occupancy(*specs[-1], slot=k)
    with pytest.raises(IndexError):
        multi.eng.set_map_occupancy(*specs[0], slot=_lib.F110_MAX_MAPS)
    which = np.arange(B) % 4
    rng = np.random.default_rng(1)
    poses = np.zeros((B, 1, 3))
    poses[:, 0, :2] = rng.uniform(-3, 3, (B, 2))
    poses[:, 0, 2] = rng.uniform(-3, 3, B)
    acts = workload.action_pool(4, B, 1)
    singles = [F110VecEnv(B, map=workload.EXAMPLE_MAP, num_agents=1) for n in names]
    for sgl, spec in zip(singles, specs):
        sgl.eng.set_map_occupancy(*spec)
    # all four kinds interleaved (one launch per run of envs of a kind), then the two maps of one kind only (one launch)
    for pattern in (which, np.where(which % 2 == 0, 0, 3)):
        multi.eng.assign_maps(np.asarray(slots)[pattern])
        om = multi.reset(poses)[0]
        outs = [s.reset(poses)[0] for s in singles]
        for step in range(4):
            for k in set(pattern.tolist()):
                sel = torch.as_tensor(pattern == k, device=multi.device)
                assert torch.equal(om['scans'][sel], outs[k]['scans'][sel]), (step, k)
                assert torch.equal(om['collisions'][sel], outs[k]['collisions'][sel])
            om = multi.step(acts[step])[0]
            outs = [s.step(acts[step])[0] for s in singles]
    assert not torch.equal(outs[0]['scans'][0], outs[3]['scans'][0])
    for e in [multi] + singles:
        e.close()


def test_randomize_tracks_blocks_drive_their_own_track():
    from red_gym_amd import F110VecEnv, workload
    B = 128
    env = F110VecEnv(B, map=workload.EXAMPLE_MAP, num_agents=1, autoreset=False)
    tracks, assign = env.randomize_tracks([11, 12, 13, 14])
    assert len(tracks) == 4 and assign.tolist() == sorted(assign.tolist()) and set(assign) == {0, 1, 2, 3}
    poses = np.zeros((B, 1, 3))
    for e in range(B):
        wp = tracks[assign[e]].waypoints
        poses[e, 0] = wp[(e * 5) % len(wp)]
    obs = env.reset(poses)[0]
    assert not bool(obs['collisions'].any())
    # on its own track a car sits mid-corridor: left/right beams see the walls ~1.7 m away (corners excepted)
    side = obs['scans'][:, 0, [180, 900]]
    assert float(side.min()) > 0.8 and 1.4 < float(side.median()) < 2.2 and float((side > 4.0).float().mean()) < 0.1
    # closed loop on four tracks at once: one planner launch per block of envs
    wps = [torch.as_tensor(np.column_stack([t.waypoints[:, :2], np.full(len(t.waypoints), 3.0)]), device=env.device) for t in tracks]
    crashed = torch.zeros(B, dtype=torch.bool, device=env.device)
    for _ in range(300):
        obs = env.step(env.pure_pursuit_blocks(wps, assign, 1.5, 1.0))[0]
        crashed |= obs['collisions'][:, 0] > 0
    assert int(crashed.sum()) == 0
    moved = torch.linalg.norm(env.state[:, 0, :2] - torch.as_tensor(poses[:, 0, :2], device=env.device), dim=1)
    assert float(moved.median()) > 4.0
    env.close()
