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
    # error behaviour: unused slot, odd split of a workgroup
    with pytest.raises(IndexError):
        multi.eng.assign_maps(np.full(B, 5))
    bad = np.zeros(B, dtype=np.int32); bad[1:] = 1
    with pytest.raises(ValueError):
        multi.eng.assign_maps(bad)
    multi.eng.assign_maps(None)
    assert torch.equal(multi.reset(poses)[0]['scans'], singles[0].reset(poses)[0]['scans'])
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
