"""GPU parity of the scan -> bitmap rasteriser against oracle/lidar_bitmap.c through the C ABI.
The oracle restates OpenCV 4.11's drawing code (cv2 is absent: parity unpinned, see the oracle's header);
the bar here is bit-exact images between the kernel's parallel formulation and that sequential restatement."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scans(n, nb=1080, seed=0, kind='track'):
    rng = np.random.default_rng(seed)
    th = np.linspace(0, 2 * np.pi, nb)
    out = np.empty((n, nb))
    for i in range(n):
        if kind == 'track':      # corridor-like, with occlusion jumps and noise
            base = 2.0 + 1.5 * np.abs(np.sin(th * rng.integers(1, 4) + rng.uniform(0, 6)))
            base = base / np.maximum(np.abs(np.cos(th + rng.uniform(0, 6))), 0.08)
            jumps = rng.random(nb) < 0.01
            base = np.where(np.cumsum(jumps) % 2 == 1, base * rng.uniform(1.5, 4), base)
            out[i] = np.clip(base, 0, 30) + rng.normal(0, 0.01, nb)
        elif kind == 'noise':    # every beam independent: maximally jagged polygon
            out[i] = rng.uniform(0, rng.choice([3.0, 12.0, 30.0]), nb)
        elif kind == 'far':      # everything outside the image
            out[i] = rng.uniform(20, 30, nb)
        else:                    # tiny: all points in a few pixels around the centre
            out[i] = rng.uniform(0, 0.3, nb)
    return out


@pytest.mark.parametrize('mode', ['FILL', 'POLYGON', 'RAYS'])
@pytest.mark.parametrize('kind', ['track', 'noise', 'far', 'tiny'])
def test_bitmap_matches_oracle(mode, kind):
    from oracle import bitmap as ob
    from red_gym_amd.lidar import LidarBitmap
    scans = _scans(24, seed=17 * ['FILL', 'POLYGON', 'RAYS'].index(mode) + len(kind), kind=kind)
    kw = dict(bg_color='black', draw_mode=mode, target_beam_count=600 if mode != 'RAYS' else 50)
    want = ob.lidar_to_bitmap(scans, **kw)
    r = LidarBitmap(1080, **kw)
    got = r(torch.as_tensor(scans, device='cuda')).cpu().numpy()
    bad = np.argwhere(got != want)
    assert bad.size == 0, 'first mismatches (img, row, col): %s' % bad[:5].tolist()
    r.close()


@pytest.mark.parametrize('opts', [
    dict(winding_dir='CW', bg_color='white', draw_mode='FILL', channels=3),
    dict(draw_mode='FILL', channels=4, fov=4.7, starting_angle=0.3, draw_center=False),
    dict(draw_mode='FILL', output_image_dims=(100, 100), max_scan_radius=12.0, scaling_factor=None),
    dict(draw_mode='FILL', output_image_dims=(96, 200), target_beam_count=1079),
    dict(draw_mode='POLYGON', output_image_dims=(33, 47), target_beam_count=7),
    dict(draw_mode='RAYS', output_image_dims=(64, 64), channels=3, target_beam_count=20, scaling_factor=3),
    dict(draw_mode='FILL', output_image_dims=(512, 512), scaling_factor=40, bg_color='black'),
    dict(draw_mode='FILL', colors=(0, 180)),          # src/bitmap.py:60 grey levels
])
def test_bitmap_options_match_oracle(opts):
    from oracle import bitmap as ob
    from red_gym_amd.lidar import LidarBitmap
    scans = np.concatenate([_scans(6, seed=5, kind='track'), _scans(4, seed=6, kind='noise')])
    want = ob.lidar_to_bitmap(scans, **opts)
    r = LidarBitmap(1080, **opts)
    got = r(torch.as_tensor(scans, device='cuda')).cpu().numpy()
    assert got.shape == want.shape and got.dtype == np.uint8
    assert np.array_equal(got, want)
    # f32 scans (the step path's own scan buffer) are widened exactly
    s32 = scans.astype(np.float32)
    assert np.array_equal(r(torch.as_tensor(s32, device='cuda')).cpu().numpy(), ob.lidar_to_bitmap(s32.astype(np.float64), **opts))
    r.close()


def test_reference_signature_single_scan_and_asserts():
    from oracle import bitmap as ob
    from weap_util.lidar import lidar_to_bitmap
    scan = _scans(1, seed=3)[0]
    img = lidar_to_bitmap(scan, output_image_dims=(256, 256), bg_color='black', draw_mode='FILL')   # src/SAL.py:76
    assert isinstance(img, np.ndarray) and img.shape == (256, 256) and img.dtype == np.uint8
    assert np.array_equal(img, ob.lidar_to_bitmap(scan, bg_color='black', draw_mode='FILL'))
    blind = lidar_to_bitmap(scan=scan, channels=3, fov=2 * np.pi, target_beam_count=50, draw_mode='RAYS', bg_color='black')
    assert blind.shape == (256, 256, 3)                                                              # lidar_example.py:104
    assert np.array_equal(blind, ob.lidar_to_bitmap(scan, channels=3, target_beam_count=50, draw_mode='RAYS', bg_color='black'))
    with pytest.raises(AssertionError):
        lidar_to_bitmap(scan, channels=2)
    with pytest.raises(AssertionError):
        lidar_to_bitmap(scan, target_beam_count=1080)
    with pytest.raises(AssertionError):
        lidar_to_bitmap(scan, fov=7.0)
    with pytest.raises(ValueError):
        lidar_to_bitmap(scan, scaling_factor=None)


def test_bitmap_of_step_scans_and_properties():
    """Bitmaps of the env's own scans: FILL covers POLYGON's outline (outside the centre marker), the
    centre marker is background, and the first images equal the oracle's."""
    from oracle import bitmap as ob
    from red_gym_amd import F110VecEnv, workload
    from red_gym_amd.lidar import LidarBitmap
    B = 512
    env = F110VecEnv(B, map=workload.EXAMPLE_MAP, num_agents=1, autoreset=True)
    env.reset(workload.spawn_poses(B, 1))
    obs = env.step(workload.action_pool(1, B, 1)[0])[0]
    scans = obs['scans'][:, 0]                     # [B, 1080] f32 view, row stride 1080
    fill = LidarBitmap(1080, bg_color='black', draw_mode='FILL')
    poly = LidarBitmap(1080, bg_color='black', draw_mode='POLYGON', draw_center=False)
    f, p = fill(scans), poly(scans)
    assert bool(((p > 0) <= ((f > 0) | _center_mask(f.shape[-2:], f.device))).all())
    assert int(f[:, 126:131, 126:131].max()) == 0
    want = ob.lidar_to_bitmap(scans[:32].double().cpu().numpy(), bg_color='black', draw_mode='FILL')
    assert np.array_equal(f[:32].cpu().numpy(), want)
    assert 0.02 < float((f > 0).float().mean()) < 0.9
    env.close(); fill.close(); poly.close()


def _center_mask(shape, device):
    m = torch.zeros(shape, dtype=torch.bool, device=device)
    m[shape[0] // 2 - 2:shape[0] // 2 + 3, shape[1] // 2 - 2:shape[1] // 2 + 3] = True
    return m


def test_scan_occupancy_matches_oracle_and_dataset_shape():
    from oracle import bitmap as ob
    from red_gym_amd.lidar import scan_occupancy
    scans = np.concatenate([_scans(8, seed=11, kind='track'), _scans(4, seed=12, kind='tiny'), _scans(4, seed=13, kind='far')])
    got = scan_occupancy(torch.as_tensor(scans, device='cuda')).cpu().numpy()
    want = np.stack([ob.occupancy(s) for s in scans])
    assert got.shape == (16, 256, 256) and got.dtype == np.uint8 and set(np.unique(got)) <= {0, 1}
    assert np.array_equal(got, want)


@pytest.mark.parametrize('mode,T,dims', [('FILL', 600, (256, 256)), ('FILL', 1079, (256, 256)), ('FILL', 2047, (512, 320)),
                                         ('POLYGON', 600, (256, 256)), ('RAYS', 50, (256, 256)), ('FILL', 100, (97, 61))])
def test_workgroups_that_draw_many_images(mode, T, dims, monkeypatch):
    """bitmap_kernel's workgroups loop over images and fetch the ranges two images ahead into LDS (global_load_lds), issued
    by half of the waves while the others work.  A launch of 3 workgroups for 41 images makes every workgroup draw 13-14 of
    them (F110_BM_GRID: the launch's workgroup count; normally one per resident slot): every image, the first, the prefetched
    ones and the last two (nothing left to prefetch), `==` the oracle's -- fp64 scans (two loads per range) and fp32."""
    from oracle import bitmap as ob
    from red_gym_amd.lidar import LidarBitmap
    nb = 2048 if T > 1079 else 1080
    scans = np.concatenate([_scans(17, nb=nb, seed=31, kind='track'), _scans(12, nb=nb, seed=32, kind='noise'),
                            _scans(6, nb=nb, seed=33, kind='far'), _scans(6, nb=nb, seed=34, kind='tiny')])
    scans = scans[np.random.default_rng(5).permutation(len(scans))]
    kw = dict(bg_color='black', draw_mode=mode, target_beam_count=T, output_image_dims=dims)
    want = ob.lidar_to_bitmap(scans, **kw)
    r = LidarBitmap(nb, **kw)
    for grid in ('3', '7', '41', None):
        if grid is None:
            monkeypatch.delenv('F110_BM_GRID', raising=False)
        else:
            monkeypatch.setenv('F110_BM_GRID', grid)
        got = r(torch.as_tensor(scans, device='cuda')).cpu().numpy()
        bad = np.argwhere(got != want)
        assert bad.size == 0, (grid, 'first mismatches (img, row, col): %s' % bad[:5].tolist())
        s32 = scans.astype(np.float32)
        got32 = r(torch.as_tensor(s32, device='cuda')).cpu().numpy()
        assert np.array_equal(got32, ob.lidar_to_bitmap(s32.astype(np.float64), **kw)), grid
    r.close()


def test_bitmap_full_batch_finishes_and_is_deterministic():
    from red_gym_amd.lidar import LidarBitmap
    n = 8192
    scans = torch.as_tensor(_scans(64, seed=21), device='cuda', dtype=torch.float32).repeat(n // 64, 1)
    r = LidarBitmap(1080, bg_color='black', draw_mode='FILL')
    a = r(scans)
    b = r(scans)
    assert torch.equal(a, b) and torch.equal(a[:64], a[64:128])   # LDS atomics are order-independent
    r.close()


def test_bitmap_fuzz_random_options():
    """Random image sizes, beam counts, scales, start angles and modes, scans with every point in / out of the image."""
    from oracle import bitmap as ob
    from red_gym_amd.lidar import LidarBitmap
    rng = np.random.default_rng(2024)
    for trial in range(40):
        nb = int(rng.choice([64, 271, 1080, 2000]))
        T = int(rng.integers(3, min(nb, 700)))
        dims = (int(rng.integers(5, 300)), int(rng.integers(5, 300)))
        if trial % 3 == 0:
            dims = (int(rng.choice([64, 128, 256])),) * 2        # the vector-store path (cols % 16 == 0)
        opts = dict(winding_dir=str(rng.choice(['CW', 'CCW'])), starting_angle=float(rng.uniform(-3.2, 3.2)),
                    scaling_factor=float(rng.choice([0.5, 3.0, 10.0, 40.0])), bg_color=str(rng.choice(['black', 'white'])),
                    draw_center=bool(rng.integers(2)), output_image_dims=dims, target_beam_count=T,
                    fov=float(rng.uniform(0.5, 2 * np.pi)), draw_mode=str(rng.choice(['FILL', 'POLYGON', 'RAYS'])),
                    channels=int(rng.choice([1, 3, 4])))
        scans = rng.uniform(0, float(rng.choice([0.5, 5.0, 30.0])), (5, nb))
        scans[0] = 0.0                                           # degenerate: every point on the centre
        scans[1, ::7] *= 10                                      # spikes far outside the image
        want = ob.lidar_to_bitmap(scans, **opts)
        r = LidarBitmap(nb, **opts)
        got = r(torch.as_tensor(scans, device='cuda')).cpu().numpy()
        assert np.array_equal(got, want), (trial, opts)
        r.close()


def test_point_stage_equals_reference_draw_calls(golden):
    """The kernel's point stage (f110_bitmap_points, the same device function bitmap_kernel starts with) against
    the integers the reference handed to cv2.fillPoly / polylines / line (g10: weap_util/weap_util/lidar.py run with
    a recording cv2): beam subset, angle table, scaling, rint -- `==` for 48 scans x 8 option sets, f64 scans.
    The rasterisation of those points stays pinned only to the OpenCV restatement (parity unpinned)."""
    import json
    from red_gym_amd.lidar import LidarBitmap
    g = golden('g10_bitmap_calls.npz')
    grid = json.loads(str(g['options']))
    scans = torch.as_tensor(g['scans'], device='cuda')
    off, val = g['arg_off'], g['arg_val']
    for oi, opt in enumerate(grid):
        r = LidarBitmap(1080, **opt)
        pts = r.points(scans).cpu().numpy()
        T = opt.get('target_beam_count', 600)
        sel = np.flatnonzero(g['opt_id'] == oi)
        seen = 0
        for si in range(scans.shape[0]):
            calls = sel[g['scan_id'][sel] == si]
            if opt['draw_mode'] == 'RAYS':
                lines = [i for i in calls if g['op'][i] == 2]
                ref = np.array([val[off[i] + 2:off[i] + 4] for i in lines])    # cv2.line(center, p)
            else:
                i = calls[0]
                assert g['op'][i] in (0, 1)
                ref = val[off[i]:off[i + 1]].reshape(-1, 2)
            assert ref.shape == (T, 2) and np.array_equal(pts[si], ref), (oi, si)
            seen += 1
        assert seen == scans.shape[0]
        r.close()


def test_occupancy_kernel_equals_reference_dataset(golden):
    """occupancy_kernel against the dataset the reference's f1tenth_gym/examples/lidar.py main() produced in the
    dev container (g12: its own scans, its own (N,256,256) uint8 array): `==`, f64 and f32 scans."""
    from red_gym_amd.lidar import scan_occupancy
    g = golden('g12_pointgrid.npz')
    shape = tuple(g['shape'])
    data = np.unpackbits(g['data_bits'], axis=-1)[..., :shape[-1]].reshape(shape)
    scans = torch.as_tensor(g['scans'], device='cuda')
    got = scan_occupancy(scans).cpu().numpy()
    assert got.dtype == np.uint8 and np.array_equal(got, data)
    # fp32 scans (what f110_step writes): identical unless a point sits within fp32 rounding of a cell edge
    got32 = scan_occupancy(scans.float()).cpu().numpy()
    assert (got32 != data).mean() < 1e-4
