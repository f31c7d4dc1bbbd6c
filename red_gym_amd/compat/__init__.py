"""Minimal stand-ins for third-party packages the reference's CALLERS import but this
image lacks (`gym`, `numba`, `pyglet.gl`), so that scripts written against the reference
(examples/waypoint_follow.py-style) run unchanged.  Nothing here is used when the real
package is importable.

    from red_gym_amd import compat; compat.install_missing()
"""
import importlib
import sys
import types


def _have(name):
    try:
        importlib.import_module(name)
        return True
    except Exception:
        return False


def _make_gym():
    gym = types.ModuleType('gym')
    registry = {}

    class Env(object):
        metadata = {}

    def register(id, entry_point, **kw):  # noqa: A002
        registry[id] = entry_point

    def make(id, **kwargs):  # noqa: A002
        """gym 0.19 semantics used by the callers: 'pkg:env-id' imports pkg first."""
        if ':' in id:
            pkg, id = id.split(':', 1)
            importlib.import_module(pkg)
        if id not in registry:
            raise KeyError('No registered env with id: %s' % id)
        mod_name, cls_name = registry[id].split(':')
        cls = getattr(importlib.import_module(mod_name), cls_name)
        return cls(**kwargs)

    gym.Env, gym.make, gym.register = Env, make, register
    envs = types.ModuleType('gym.envs')
    registration = types.ModuleType('gym.envs.registration')
    registration.register = register
    envs.registration = registration
    gym.envs = envs
    for sub in ('error', 'spaces', 'utils'):
        m = types.ModuleType('gym.' + sub)
        setattr(gym, sub, m)
        sys.modules['gym.' + sub] = m
    seeding = types.ModuleType('gym.utils.seeding')
    gym.utils.seeding = seeding
    sys.modules.update({'gym': gym, 'gym.envs': envs, 'gym.envs.registration': registration,
                        'gym.utils.seeding': seeding})
    return gym


def install_missing():
    installed = []
    if not _have('gym'):
        _make_gym()
        installed.append('gym')
        if 'f110_gym' in sys.modules:  # registration was skipped at import time
            sys.modules['gym'].register(id='f110-v0', entry_point='f110_gym.envs:F110Env')
    if not _have('numba'):
        nb = types.ModuleType('numba')

        def njit(*args, **kwargs):
            if len(args) == 1 and callable(args[0]) and not kwargs:
                return args[0]
            return lambda f: f
        nb.njit = njit
        sys.modules['numba'] = nb
        installed.append('numba')
    if not _have('pyglet'):
        pyglet = types.ModuleType('pyglet')
        pyglet.options = {}
        gl = types.ModuleType('pyglet.gl')
        gl.GL_POINTS = 0
        pyglet.gl = gl
        sys.modules['pyglet'] = pyglet
        sys.modules['pyglet.gl'] = gl
        installed.append('pyglet')
    return installed
