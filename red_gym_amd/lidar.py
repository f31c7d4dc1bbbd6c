"""Scan -> bird's-eye bitmap on the GPU: the host-side mirror of the reference's
weap_util/weap_util/lidar.py:105-154 `lidar_to_bitmap` (identical bodies in src/SAL.py:274-395 and
src/bitmap.py:4-140), which both RL callers run on every step's scan (src/SAL.py:76,119,
examples/lidar_example.py:104-105).

`lidar_to_bitmap(scan, ...)` keeps the reference's name, arguments, assertions and return value for one
scan; `LidarBitmap` is the batched form ([N, num_beams] device tensor -> [N, H, W(, C)] uint8 device
tensor, one workgroup per image).  Both run libf110_hip.so's bitmap_kernel; there is no CPU path.

The reference draws with OpenCV 4.11 (weap_util/setup.py:8).  cv2 is not available to this build, so the
kernel is checked against a restatement of OpenCV's algorithms (parity unpinned).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

DRAW_MODES = {'FILL': 0, 'POLYGON': 1, 'RAYS': 2}


def beam_tables(num_beams, target_beam_count, winding_dir='CCW', starting_angle=-np.pi / 2, fov=2 * np.pi):
    """lidar.py:63-72, verbatim numpy expressions: the beam subset and cos / sin of the drawing angles."""
    direction = 1 if winding_dir == 'CCW' else -1
    indices = np.linspace(0, num_beams - 1, target_beam_count, dtype=int)
    angles = starting_angle + direction * fov * np.linspace(0, 1, target_beam_count)
    return (np.ascontiguousarray(indices, np.int32), np.ascontiguousarray(np.cos(angles)),
            np.ascontiguousarray(np.sin(angles)))


class LidarBitmap:
    """Batched renderer for one fixed set of drawing options (the reference's keyword arguments)."""

    def __init__(self, num_beams, winding_dir='CCW', starting_angle=-np.pi / 2, max_scan_radius=None,
                 scaling_factor=10, bg_color='white', draw_center=True, output_image_dims=(256, 256),
                 target_beam_count=600, fov=2 * np.pi, draw_mode='POLYGON', channels=1, colors=None, device=0):
        # lidar.py:50-61,137 (the reference's assertions, same messages)
        assert channels in [1, 3, 4], "channels must 1, 3, or 4"
        assert winding_dir in ['CW', 'CCW'], "winding_dir must be either clockwise or counterclockwise"
        assert bg_color in ['black', 'white']
        assert draw_mode in ['RAYS', 'POLYGON', 'FILL']
        assert len(output_image_dims) == 2
        assert all([x > 0 for x in output_image_dims]), "output_image_dims must be at least 1x1"
        assert 0 < target_beam_count < num_beams
        assert 0 < fov <= 2 * np.pi, "FOV must be between 0 and 2pi"
        if max_scan_radius is not None:
            scaling_factor = min(output_image_dims) / max_scan_radius
        elif scaling_factor is None:
            raise ValueError("Must provide either max_scan_radius or scaling_factor")
        bg, draw = colors if colors is not None else ((0, 255) if bg_color == 'black' else (255, 0))
        self.lib = _lib.load()
        self.device = torch.device('cuda', device) if isinstance(device, int) else torch.device(device)
        self.num_beams, self.channels = int(num_beams), int(channels)
        self.target_beam_count = int(target_beam_count)
        self.rows, self.cols = int(output_image_dims[0]), int(output_image_dims[1])
        idx, cs, sn = beam_tables(num_beams, target_beam_count, winding_dir, starting_angle, fov)
        cfg = _lib.BitmapConfig(device=self.device.index or 0, num_beams=num_beams,
                                target_beam_count=target_beam_count, rows=self.rows, cols=self.cols,
                                channels=channels, draw_mode=DRAW_MODES[draw_mode], bg_value=int(bg),
                                draw_value=int(draw), draw_center=int(bool(draw_center)),
                                scaling_factor=float(scaling_factor))
        h = C.c_void_p()
        _lib.check(self.lib.f110_bitmap_create(C.byref(cfg), idx.ctypes.data, cs.ctypes.data, sn.ctypes.data,
                                               C.byref(h)))
        self.h = h

    def __call__(self, scans, out=None):
        """scans: [N, num_beams] (or [num_beams]) float32 / float64 tensor on the renderer's device."""
        if not torch.is_tensor(scans):
            scans = torch.as_tensor(np.asarray(scans, dtype=np.float64), device=self.device)
        single = scans.dim() == 1
        s = scans.reshape(-1, scans.shape[-1])
        if s.dtype not in (torch.float32, torch.float64):
            s = s.to(torch.float64)
        if s.device != self.device:
            s = s.to(self.device)
        if s.stride(-1) != 1:
            s = s.contiguous()
        assert s.shape[-1] == self.num_beams, 'scan length %d != %d' % (s.shape[-1], self.num_beams)
        n = s.shape[0]
        shape = (n, self.rows, self.cols) + ((self.channels,) if self.channels > 1 else ())
        if out is None:
            out = torch.empty(shape, dtype=torch.uint8, device=self.device)
        assert out.is_contiguous() and out.dtype == torch.uint8 and tuple(out.shape) == shape
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):  # the library launches on the current device's stream (f110_hip.h, conventions)
            _lib.check(self.lib.f110_bitmap_render(self.h, s.data_ptr(), int(s.dtype == torch.float64), n,
                                                   s.stride(0) if n > 1 else self.num_beams, out.data_ptr(), stream))
        out = out.reshape(scans.shape[:-1] + shape[1:]) if not single else out[0]
        return out

    def points(self, scans):
        """The integer points of lidar.py:63-73 (what the reference passes to the cv2 draw calls):
        [N, target_beam_count, 2] int32 device tensor of (x, y)."""
        s = scans.reshape(-1, scans.shape[-1])
        if s.dtype not in (torch.float32, torch.float64):
            s = s.to(torch.float64)
        s = s.to(self.device).contiguous()
        n = s.shape[0]
        out = torch.empty((n, self.target_beam_count, 2), dtype=torch.int32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            _lib.check(self.lib.f110_bitmap_points(self.h, s.data_ptr(), int(s.dtype == torch.float64), n, s.stride(0),
                                                   out.data_ptr(), stream))
        torch.cuda.current_stream(self.device).synchronize()  # `s` may be a temporary
        return out

    def close(self):
        if getattr(self, 'h', None) is not None and self.h:
            torch.cuda.synchronize(self.device)
            self.lib.f110_bitmap_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_cache = {}


def lidar_to_bitmap(scan, winding_dir='CCW', starting_angle=-np.pi / 2, max_scan_radius=None, scaling_factor=10,
                    bg_color='white', draw_center=True, output_image_dims=(256, 256), target_beam_count=600,
                    fov=2 * np.pi, draw_mode='POLYGON', channels=1):
    """weap_util.lidar.lidar_to_bitmap (lidar.py:105-154): one scan -> np.ndarray uint8 [H, W(, C)].
    A device tensor scan (or a [N, num_beams] batch) returns a device tensor instead."""
    n_beams = int(scan.shape[-1]) if hasattr(scan, 'shape') else len(scan)
    key = (n_beams, winding_dir, float(starting_angle), max_scan_radius, scaling_factor, bg_color, bool(draw_center),
           tuple(output_image_dims), target_beam_count, float(fov), draw_mode, channels)
    r = _cache.get(key)
    if r is None:
        if len(_cache) >= 16:
            _cache.pop(next(iter(_cache))).close()
        r = _cache[key] = LidarBitmap(n_beams, winding_dir, starting_angle, max_scan_radius, scaling_factor, bg_color,
                                      draw_center, output_image_dims, target_beam_count, fov, draw_mode, channels)
    if torch.is_tensor(scan) and scan.is_cuda:
        return r(scan)
    return r(torch.as_tensor(np.asarray(scan, dtype=np.float64), device=r.device)).cpu().numpy()


def occupancy_tables(num_beams):
    """f1tenth_gym/examples/lidar.py:215: angles = np.linspace(-135, 135, n) * pi / 180."""
    angles = np.linspace(-135, 135, num_beams) * np.pi / 180.0
    return np.cos(angles), np.sin(angles)


def scan_occupancy(scans, max_range=30.0, lo=-10.0, hi=10.0, grid_size=256):
    """Point-occupancy grids of f1tenth_gym/examples/lidar.py:212-244 (the frames of the reference's
    lidar_datasets/*.npz): scans [N, num_beams] device tensor -> uint8 [N, grid, grid] of 0 / 1."""
    lib = _lib.load()
    single = scans.dim() == 1
    s = scans.reshape(-1, scans.shape[-1])
    if s.dtype not in (torch.float32, torch.float64):
        s = s.to(torch.float64)
    if s.stride(-1) != 1:
        s = s.contiguous()
    n, nb = s.shape
    cs, sn = (torch.as_tensor(t, device=s.device) for t in occupancy_tables(nb))
    out = torch.empty((n, grid_size, grid_size), dtype=torch.uint8, device=s.device)
    stream = torch.cuda.current_stream(s.device).cuda_stream
    with torch.cuda.device(s.device):
        _lib.check(lib.f110_scan_occupancy(s.data_ptr(), int(s.dtype == torch.float64), n, s.stride(0) if n > 1 else nb, nb,
                                           cs.data_ptr(), sn.data_ptr(), max_range, lo, hi, grid_size, out.data_ptr(),
                                           stream))
    torch.cuda.current_stream(s.device).synchronize()  # cs / sn are temporaries
    return out[0] if single else out
