"""Dev-container-only loader for the reference's hot-path modules.

Used ONLY by make_golden.py to produce the .npz fixtures in this directory.
Nothing here is imported by tests, bench.py or the product; /root/reference
does not exist on the GPU box.

The reference's kernels are plain Python decorated with numba.njit; numba is
not installed here, so an identity `njit` stand-in is registered before import
(same source, same IEEE-754 double operations).
The package __init__ (which imports `gym`) is bypassed by pre-seeding
sys.modules with namespace modules whose __path__ points at the reference.
"""
import importlib
import sys
import types
import warnings

REF_ROOT = '/root/reference'
REF_PKG = REF_ROOT + '/gym/f110_gym'


def _identity_njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda f: f


def load():
    """Returns (laser_models, dynamic_models, collision_models, base_classes)."""
    if 'numba' not in sys.modules:
        nb = types.ModuleType('numba')
        nb.njit = _identity_njit
        sys.modules['numba'] = nb
    pkg = types.ModuleType('f110_gym')
    pkg.__path__ = [REF_PKG]
    sub = types.ModuleType('f110_gym.envs')
    sub.__path__ = [REF_PKG + '/envs']
    sys.modules['f110_gym'] = pkg
    sys.modules['f110_gym.envs'] = sub
    warnings.filterwarnings('ignore', message='Chosen integrator is RK4')
    lm = importlib.import_module('f110_gym.envs.laser_models')
    dm = importlib.import_module('f110_gym.envs.dynamic_models')
    cm = importlib.import_module('f110_gym.envs.collision_models')
    bc = importlib.import_module('f110_gym.envs.base_classes')
    return lm, dm, cm, bc


def load_env():
    """Additionally imports f110_env.F110Env with minimal `gym`/`pyglet`
    stand-ins (third-party packages absent from this image)."""
    lm, dm, cm, bc = load()
    if 'gym' not in sys.modules:
        gym = types.ModuleType('gym')

        class Env(object):
            pass
        gym.Env = Env
        for name in ('error', 'spaces', 'utils'):
            m = types.ModuleType('gym.' + name)
            setattr(gym, name, m)
            sys.modules['gym.' + name] = m
        seeding = types.ModuleType('gym.utils.seeding')
        gym.utils.seeding = seeding
        sys.modules['gym.utils.seeding'] = seeding
        sys.modules['gym'] = gym
    if 'pyglet' not in sys.modules:
        pyglet = types.ModuleType('pyglet')
        pyglet.options = {}
        gl = types.ModuleType('pyglet.gl')
        gl.GL_POINTS = 0
        pyglet.gl = gl
        sys.modules['pyglet'] = pyglet
        sys.modules['pyglet.gl'] = gl
    fe = importlib.import_module('f110_gym.envs.f110_env')
    return lm, dm, cm, bc, fe
