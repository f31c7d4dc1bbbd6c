"""MI355X-native batched F1TENTH environment (the F110Env.step hot path of
WE-Autopilot/red_gym as hand-written HIP kernels behind a C ABI).

    from red_gym_amd import F110VecEnv, F110Env, Integrator
"""
from .base_classes import Integrator  # noqa: F401


def __getattr__(name):
    # heavy imports (torch) only when the env classes are actually requested
    if name in ('F110VecEnv', 'VecObs'):
        from . import vec_env
        return getattr(vec_env, name)
    if name == 'F110Env':
        from .f110_env import F110Env
        return F110Env
    if name in ('lidar_to_bitmap', 'LidarBitmap', 'scan_occupancy'):
        from . import lidar
        return getattr(lidar, name)
    raise AttributeError(name)
