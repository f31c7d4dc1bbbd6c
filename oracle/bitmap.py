"""CPU oracle for the scan -> bitmap rasteriser -- TEST INFRASTRUCTURE, NOT PRODUCT.

ctypes front-end of oracle/lidar_bitmap.c (restates weap_util/weap_util/lidar.py:4-154 over a
restatement of OpenCV 4.11.0's drawing.cpp; PARITY UNPINNED, see that file's header).  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, 'lidar_bitmap.c')
_BUILD = os.path.join(_HERE, '_build')
_SO = os.path.join(_BUILD, 'liblidar_bitmap_oracle.so')
MODES = {'FILL': 0, 'POLYGON': 1, 'RAYS': 2}
_lib = None


def build(force=False):
    os.makedirs(_BUILD, exist_ok=True)
    if not force and os.path.exists(_SO) and os.path.getmtime(_SO) >= os.path.getmtime(_SRC):
        return _SO
    subprocess.run(['gcc', '-O2', '-std=c11', '-fPIC', '-shared', '-ffp-contract=off', '-fno-fast-math',
                    '-fopenmp', '-o', _SO, _SRC, '-lm'], check=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def tables(num_beams, target_beam_count=600, winding_dir='CCW', starting_angle=-np.pi / 2, fov=2 * np.pi):
    """lidar.py:63-72 -- the beam subset and the drawing angles (host numpy, exactly the reference's expressions)."""
    direction = 1 if winding_dir == 'CCW' else -1
    indices = np.linspace(0, num_beams - 1, target_beam_count, dtype=int)
    angles = starting_angle + direction * fov * np.linspace(0, 1, target_beam_count)
    return indices.astype(np.int32), np.cos(angles), np.sin(angles)


def lidar_to_bitmap(scan, winding_dir='CCW', starting_angle=-np.pi / 2, max_scan_radius=None, scaling_factor=10,
                    bg_color='white', draw_center=True, output_image_dims=(256, 256), target_beam_count=600,
                    fov=2 * np.pi, draw_mode='POLYGON', channels=1, colors=None):
    """Same signature as weap_util.lidar.lidar_to_bitmap (lidar.py:105-154); scan may be [n_beams] or [N, n_beams]."""
    assert channels in [1, 3, 4], "channels must 1, 3, or 4"
    assert winding_dir in ['CW', 'CCW'] and bg_color in ['black', 'white'] and draw_mode in MODES
    scans = np.ascontiguousarray(np.atleast_2d(np.asarray(scan, dtype=np.float64)))
    n, nb = scans.shape
    assert 0 < target_beam_count < nb and 0 < fov <= 2 * np.pi
    if max_scan_radius is not None:
        scaling_factor = min(output_image_dims) / max_scan_radius
    elif scaling_factor is None:
        raise ValueError("Must provide either max_scan_radius or scaling_factor")
    bg, draw = colors if colors is not None else ((0, 255) if bg_color == 'black' else (255, 0))
    idx, cs, sn = tables(nb, target_beam_count, winding_dir, starting_angle, fov)
    rows, cols = output_image_dims
    out = np.empty((n, rows, cols), np.uint8)
    lib().lidar_bitmap_batch(_p(scans), C.c_int64(n), C.c_int(nb), _p(idx), _p(cs), _p(sn),
                             C.c_int(target_beam_count), C.c_int(rows), C.c_int(cols),
                             C.c_double(scaling_factor), C.c_int(MODES[draw_mode]), C.c_int(bg), C.c_int(draw),
                             C.c_int(bool(draw_center)), _p(out))
    if channels == 3:
        out = np.stack([out] * 3, axis=-1)
    elif channels == 4:
        out = np.stack([out, out, out, np.full_like(out, 255)], axis=-1)
    return out[0] if np.ndim(scan) == 1 else out


def points(scan, num_target=600, scale=10.0, dims=(256, 256), **kw):
    scan = np.ascontiguousarray(scan, np.float64)
    idx, cs, sn = tables(scan.shape[0], num_target, **kw)
    pts = np.empty((num_target, 2), np.int64)
    lib().lidar_points(_p(scan), _p(idx), _p(cs), _p(sn), C.c_int(num_target), C.c_double(dims[0] // 2),
                       C.c_double(dims[1] // 2), C.c_double(scale), _p(pts))
    return pts


def _img(shape, fill):
    return np.full(shape, fill, np.uint8)


def fill_poly(shape, pts, color=255, bg=0):
    img = _img(shape, bg)
    pts = np.ascontiguousarray(pts, np.int64)
    lib().cv_fill_poly(_p(img), C.c_int(shape[0]), C.c_int(shape[1]), _p(pts), C.c_int(len(pts)), C.c_int(color))
    return img


def polylines(shape, pts, closed=True, color=255, bg=0):
    img = _img(shape, bg)
    pts = np.ascontiguousarray(pts, np.int64)
    lib().cv_polylines(_p(img), C.c_int(shape[0]), C.c_int(shape[1]), _p(pts), C.c_int(len(pts)),
                       C.c_int(bool(closed)), C.c_int(color))
    return img


def line(shape, p1, p2, color=255, bg=0):
    img = _img(shape, bg)
    lib().cv_line(_p(img), C.c_int(shape[0]), C.c_int(shape[1]), C.c_int64(p1[0]), C.c_int64(p1[1]),
                  C.c_int64(p2[0]), C.c_int64(p2[1]), C.c_int(color))
    return img


def rectangle_filled(shape, p1, p2, color=255, bg=0):
    img = _img(shape, bg)
    lib().cv_rectangle_filled(_p(img), C.c_int(shape[0]), C.c_int(shape[1]), C.c_int64(p1[0]), C.c_int64(p1[1]),
                              C.c_int64(p2[0]), C.c_int64(p2[1]), C.c_int(color))
    return img


def occupancy(scan, max_range=30.0, lo=-10.0, hi=10.0, grid_size=256):
    """f1tenth_gym/examples/lidar.py:212-244."""
    scan = np.ascontiguousarray(scan, np.float64)
    angles = np.linspace(-135, 135, len(scan)) * np.pi / 180.0
    cs, sn = np.cos(angles), np.sin(angles)
    grid = np.empty((grid_size, grid_size), np.uint8)
    lib().lidar_occupancy(_p(scan), _p(cs), _p(sn), C.c_int(len(scan)), C.c_double(max_range), C.c_double(lo),
                          C.c_double(hi), C.c_int(grid_size), _p(grid))
    return grid
