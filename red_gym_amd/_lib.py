"""ctypes binding of include/f110_hip.h (libf110_hip.so, built by red_gym_amd.build).

There is no CPU fallback: if the HIP library is missing or fails to load the
import raises, and every entry point returning a negative code raises here.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('F110_LIB') or os.path.join(_HERE, 'libf110_hip.so')  # F110_LIB: kernel-variant sweeps

F110_MAX_AGENTS = 32
F110_MAX_NOISE_SLOTS = 64
F110_MAX_MAPS = 4096
F110_NUM_PARAMS = 18
F110_RK4, F110_EULER = 1, 2
E_INVALID, E_HIP, E_NOMAP, E_INDEX, E_UNBOUND = -1, -2, -3, -4, -5

PARAM_KEYS = ['mu', 'C_Sf', 'C_Sr', 'lf', 'lr', 'h', 'm', 'I', 's_min', 's_max', 'sv_min', 'sv_max',
              'v_switch', 'a_max', 'v_min', 'v_max', 'width', 'length']


class Config(C.Structure):
    _fields_ = [('num_envs', C.c_int32), ('num_agents', C.c_int32), ('num_beams', C.c_int32),
                ('theta_dis', C.c_int32), ('integrator', C.c_int32), ('ego_idx', C.c_int32),
                ('device', C.c_int32), ('autoreset', C.c_int32), ('fov', C.c_double), ('eps', C.c_double),
                ('max_range', C.c_double), ('timestep', C.c_double), ('ttc_thresh', C.c_double),
                ('params', C.c_double * F110_NUM_PARAMS)]


BUFFER_FIELDS = ['state', 'steer_buf', 'steer_cnt', 'noise_step', 'spawn', 'start_rot', 'near_start',
                 'toggles', 'current_time', 'pending_reset', 'scans', 'scans_f64', 'pose_snap',
                 'collisions', 'collision_idx', 'in_collision', 'lap_counts', 'lap_times', 'done', 'checkpoint_done', 'lookups']


class BitmapConfig(C.Structure):
    _fields_ = [('device', C.c_int32), ('num_beams', C.c_int32), ('target_beam_count', C.c_int32),
                ('rows', C.c_int32), ('cols', C.c_int32), ('channels', C.c_int32), ('draw_mode', C.c_int32),
                ('bg_value', C.c_int32), ('draw_value', C.c_int32), ('draw_center', C.c_int32),
                ('scaling_factor', C.c_double)]


class Buffers(C.Structure):
    _fields_ = [(name, C.c_void_p) for name in BUFFER_FIELDS]


# every symbol include/f110_hip.h declares: name -> argtypes (restype int unless noted)
_VP, _I32, _I64, _D = C.c_void_p, C.c_int32, C.c_int64, C.c_double
SYMBOLS = {
    'f110_create': [C.POINTER(Config), C.POINTER(_VP)],
    'f110_destroy': [_VP],
    'f110_last_error': [],
    'f110_update_params': [_VP, _VP, _I32],
    'f110_set_tables': [_VP, _VP, _VP, _VP, _VP, _VP],
    'f110_set_map_occupancy': [_VP, _VP, _I32, _I32, _D, _D, _D, _D, _D],
    'f110_set_map_occupancy_dev': [_VP, _VP, _I32, _I32, _D, _D, _D, _D, _D],
    'f110_set_map_slot_occupancy': [_VP, _I32, _VP, _I32, _I32, _D, _D, _D, _D, _D],
    'f110_set_map_slot_occupancy_dev': [_VP, _I32, _VP, _I32, _I32, _D, _D, _D, _D, _D],
    'f110_assign_maps': [_VP, _VP],
    'f110_get_map_slot_dt': [_VP, _I32, _VP],
    'f110_track_mask': [_VP, _I32, _I32, _I32, _I32, _D, _D, _D, _D, _D, _VP, _VP],
    'f110_set_map_dt': [_VP, _VP, _I32, _I32, _D, _D, _D, _D, _D],
    'f110_get_map_dt': [_VP, _VP],
    'f110_edt_squared': [_VP, _I32, _I32, _VP],
    'f110_edt_squared_dev': [_VP, _I32, _I32, _VP, _VP],
    'f110_set_params_slots': [_VP, _VP, _I32],
    'f110_set_params_slot': [_VP, _I32, _VP, _I32],
    'f110_assign_params': [_VP, _VP],
    'f110_set_noise_table': [_VP, _VP, _I64],
    'f110_set_noise_slot': [_VP, _I32, _VP, _I64],
    'f110_set_noise_generator': [_VP, _I32, _VP, _D],
    'f110_assign_noise': [_VP, _VP],
    'f110_set_noise_per_env': [_VP, _VP, _D],
    'f110_noise_ensure': [_VP, _I64, _VP],
    'f110_noise_prefetch': [_VP, _I64],
    'f110_noise_set_floor': [_VP, _I64, _VP],
    'f110_noise_info': [_VP, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int64)],
    'f110_noise_read': [_VP, _I32, _I64, _I64, _VP],
    'f110_device_errors': [_VP, C.POINTER(C.c_uint32)],
    'f110_bind': [_VP, C.POINTER(Buffers)],
    'f110_reset': [_VP, _VP, _VP, _VP],
    'f110_step': [_VP, _VP, _VP],
    'f110_pack_env_size': [_VP],
    'f110_pack_env': [_VP, _I32, _VP, _VP],
    'f110_set_scan_stages': [_VP, C.c_char_p],
    'f110_launch_epoch': [_VP, C.POINTER(C.c_int64)],
    'f110_set_scan_order': [_VP, _VP],
    'f110_graph_create': [_VP, _VP, _I32, C.POINTER(_VP)],
    'f110_graph_launch': [_VP, _VP],
    'f110_graph_info': [_VP, C.POINTER(C.c_int32), C.c_char_p],
    'f110_graph_destroy': [_VP],
    'f110_pure_pursuit': [_VP, _VP, _I32, _D, _D, _D, _D, _VP, _I32, _VP, _VP],
    'f110_pure_pursuit_prepare': [_VP, _VP, _I32, _D, _D, _VP],
    'f110_pure_pursuit_workspace': [_I32, _I32],
    'f110_pure_pursuit_tracks': [_VP, _VP, _VP, _VP, _I32, _VP, _D, _D, _D, _D, _VP, _I32, _VP, _VP, _I32, _VP],
    'f110_profile_begin': [_VP, _I32],
    'f110_profile_every': [_VP, _I32],
    'f110_profile_end': [_VP, C.POINTER(C.c_double), C.POINTER(C.c_int32)],
    'f110_scan': [_VP, _VP, _I32, _VP, _VP, _VP, _VP],
    'f110_update_pose': [_VP, _VP, _VP, _VP, _VP, _I32, _VP],
    'f110_vehicle_dynamics': [_VP, _VP, _VP, _I32, _I32, _VP, _VP],
    'f110_get_vertices': [_VP, _VP, _I32, _VP, _VP],
    'f110_gjk_pairs': [_VP, _VP, _VP, _I32, _VP, _VP],
    'f110_collision_multiple': [_VP, _VP, _I32, _I32, _VP, _VP, _VP],
    'f110_check_ttc': [_VP, _VP, _VP, _I32, _VP, _VP],
    'f110_ray_cast': [_VP, _VP, _VP, _I32, _VP, _VP, _VP],
    'f110_check_done': [_VP, _VP, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _VP, _VP, _VP, _VP, _VP, _VP, _VP],
    'f110_bitmap_create': [C.POINTER(BitmapConfig), _VP, _VP, _VP, C.POINTER(_VP)],
    'f110_bitmap_destroy': [_VP],
    'f110_bitmap_render': [_VP, _VP, _I32, _I64, _I64, _VP, _VP],
    'f110_bitmap_points': [_VP, _VP, _I32, _I64, _I64, _VP, _VP],
    'f110_scan_occupancy': [_VP, _I32, _I64, _I64, _I32, _VP, _VP, _D, _D, _D, _I32, _VP, _VP],
}


class F110Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('f110 error %d: %s' % (code, msg))
        self.code = code


_lib = None


def load():
    """Loads libf110_hip.so; raises if it is absent (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError('%s not found: build it with `python -m red_gym_amd.build` '
                          '(hipcc --offload-arch=gfx950); there is no CPU fallback.' % LIB_PATH)
    # torch bundles its own libamdhip64 (soname libamdhip64.so.7).  It must be in the
    # process BEFORE this library is loaded so that both share ONE HIP runtime (ours then
    # binds to the already-loaded soname); the other order maps a second runtime, which
    # cannot see the device.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SYMBOLS.items():
        if os.environ.get('F110_LIB_OLDER') == '1' and not hasattr(lib, name):
            continue             # A/B runs against a library built from an earlier tree (tools/sweep.py): newer entry points are absent
        fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.f110_last_error.restype = C.c_char_p
    lib.f110_pure_pursuit_workspace.restype = C.c_int64
    lib.f110_pack_env_size.restype = C.c_int64
    lib.f110_destroy.restype = None
    lib.f110_bitmap_destroy.restype = None
    lib.f110_graph_destroy.restype = None
    _lib = lib
    return lib


def check(rc):
    """Maps C error codes to the exceptions the reference raises for the same
    conditions (ValueError base_classes.py:619 / laser_models.py:446, IndexError :527)."""
    if rc == 0:
        return
    msg = load().f110_last_error().decode('utf-8', 'replace')
    if rc in (E_INVALID, E_NOMAP):
        raise ValueError(msg)
    if rc == E_INDEX:
        raise IndexError(msg)
    raise F110Error(rc, msg)
