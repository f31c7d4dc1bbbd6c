"""Pins of the "next" rows against fixtures generated from the reference's own code
(tests/golden/make_golden_r2.py, dev container).  What each fixture pins -- and what stays unpinned -- is said
in the test; the OpenCV / shapely / matplotlib rasterisation itself is out of reach (not installed)."""
import json
import os

import numpy as np
import pytest

from oracle import bitmap as ob


def draw_calls(golden):
    """g10 as a list of (option index, scan index, op, int args, colour word) in call order."""
    g = golden('g10_bitmap_calls.npz')
    off, val = g['arg_off'], g['arg_val']
    calls = [(int(g['opt_id'][i]), int(g['scan_id'][i]), int(g['op'][i]), val[off[i]:off[i + 1]].astype(np.int64),
              int(g['color'][i])) for i in range(len(g['op']))]
    return g['scans'], json.loads(str(g['options'])), calls, g['out_shape']


def expected_calls(points, opt):
    """The cv2 calls weap_util/weap_util/lidar.py:82-100 issues for one scan, given its integer points: what a
    reader of the reference expects; the fixture checks that reading too."""
    dims = tuple(opt.get('output_image_dims', (256, 256)))
    mode = opt.get('draw_mode', 'POLYGON')
    bg, draw = (0, 255) if opt.get('bg_color', 'white') == 'black' else (255, 0)
    centre = np.array([dims[0] // 2, dims[1] // 2])
    out = []
    if mode == 'FILL':
        out.append((0, points.reshape(-1), draw))
    elif mode == 'POLYGON':
        out.append((1, points.reshape(-1), draw | 1 << 16 | 1 << 20))           # isClosed=True, thickness=1
    else:
        for p in points:
            out.append((2, np.concatenate([centre, p]), draw))
            out.append((3, np.concatenate([p - 2, p + 2]), draw | (0xff << 20)))  # thickness=-1
    if opt.get('draw_center', True):
        out.append((3, np.concatenate([centre - 2, centre + 2]), (bg if mode == 'FILL' else draw) | (0xff << 20)))
    return out


def oracle_points(scan, opt):
    dims = tuple(opt.get('output_image_dims', (256, 256)))
    scale = opt.get('scaling_factor', 10)
    if opt.get('max_scan_radius') is not None:
        scale = min(dims) / opt['max_scan_radius']
    kw = {k: opt[k] for k in ('winding_dir', 'starting_angle', 'fov') if k in opt}
    return ob.points(scan, opt.get('target_beam_count', 600), float(scale), dims, **kw)


def test_bitmap_point_stage_equals_the_reference_draw_calls(golden):
    """f2, pinned part: every integer the reference hands to cv2.fillPoly / polylines / line / rectangle (beam
    subset lidar.py:63-64, angles :67, rint of the points :70-73, centre rectangle :90-91, colours :61) for 48
    scans x 8 option sets equals what oracle/lidar_bitmap.c computes before it rasterises.  UNPINNED remains the
    rasterisation of those calls (oracle = restatement of OpenCV 4.11 drawing.cpp, cv2 not installed)."""
    scans, grid, calls, shapes = draw_calls(golden)
    by_key = {}
    for oi, si, op, args, col in calls:
        by_key.setdefault((oi, si), []).append((op, args, col))
    assert len(by_key) == len(scans) * len(grid)
    for (oi, si), got in by_key.items():
        want = expected_calls(oracle_points(scans[si], grid[oi]), grid[oi])
        assert len(got) == len(want), (oi, si)
        for (gop, gargs, gcol), (wop, wargs, wcol) in zip(got, want):
            assert gop == wop and gcol == wcol and np.array_equal(gargs, wargs), (oi, si, gop)
    # the shape the reference returns for each option set (channels :139-152)
    for oi, opt in enumerate(grid):
        dims = tuple(opt.get('output_image_dims', (256, 256)))
        ch = opt.get('channels', 1)
        want = dims + ((ch,) if ch > 1 else (0,))
        assert tuple(shapes[oi * len(scans)]) == want
        img = ob.lidar_to_bitmap(scans[0], **opt)
        assert img.shape == dims + ((ch,) if ch > 1 else ())


def test_track_centre_line_equals_the_reference_generator(golden):
    """f3: create_track() of unittest/random_trackgen.py run up to its shapely call (:56-159) for 10 seeds x 6
    consecutive calls; red_gym_amd.trackgen.create_centerline on the same RandomState stream gives the same
    outcome (gave up / centre line) and the same centre line, bit for bit.  UNPINNED remains the wall image
    (shapely offset + matplotlib stroke + cv2 re-read, :161-218), replaced by a level-set construction."""
    from red_gym_amd import trackgen as tg
    g = golden('g11_centerline.npz')
    n_lines = 0
    for seed in g['seeds']:
        status = g['seed%d_status' % seed]
        rng = np.random.RandomState(int(seed))
        j = 0
        for ok in status:
            c = tg.create_centerline(rng)
            assert (c is not None) == bool(ok), (seed, j)
            if ok:
                want = g['seed%d_line%d' % (seed, j)]
                assert c.shape == want.shape and np.array_equal(c, want), (seed, j)
                j += 1
                n_lines += 1
        # the product's retry loop returns the first closed line of the stream
        first = int(np.argmax(status))
        assert np.array_equal(tg.random_centerline(int(seed)), g['seed%d_line0' % seed]) and status[first] == 1
    assert n_lines >= 40


def test_point_grid_oracle_equals_the_reference_dataset(golden):
    """f4: main() of f1tenth_gym/examples/lidar.py run headless for 5 episodes: the (N,256,256) uint8 array it
    passes to np.savez_compressed equals oracle.bitmap.occupancy of the scans it saw (:212-244), and has the
    format of the reference's shipped lidar_datasets/*.npz (key 'data', uint8, values {0, 1})."""
    g = golden('g12_pointgrid.npz')
    shape = tuple(g['shape'])
    data = np.unpackbits(g['data_bits'], axis=-1)[..., :shape[-1]].reshape(shape)
    assert shape[1:] == (256, 256) and set(np.unique(data)) <= {0, 1} and data.sum() > 0
    for k in range(shape[0]):
        assert np.array_equal(ob.occupancy(g['scans'][k]), data[k]), k


def test_reference_dataset_files_have_the_format_the_writer_targets():
    """The reference ships datasets its recorder wrote (f1tenth_gym/examples/lidar_datasets/*.npz): key 'data',
    uint8 [N,256,256], values {0, 1} -- the format LidarDatasetWriter.save reproduces (checked on the GPU against
    g12).  Dev container only: the reference tree does not exist on the GPU box."""
    import glob
    files = sorted(glob.glob('/root/reference/f1tenth_gym/examples/lidar_datasets/*.npz'))
    if not files:
        pytest.skip('reference tree not present')
    for f in files[:3]:
        d = np.load(f, allow_pickle=False)   # plain arrays: nothing is unpickled
        assert d.files == ['data']
        a = d['data']
        assert a.dtype == np.uint8 and a.ndim == 3 and a.shape[1:] == (256, 256) and set(np.unique(a)) <= {0, 1}
