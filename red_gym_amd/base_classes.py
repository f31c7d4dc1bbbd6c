"""Names the reference exposes from f110_gym.envs.base_classes that callers import
(examples/waypoint_follow.py:2 `from f110_gym.envs.base_classes import Integrator`)."""
from enum import Enum


class Integrator(Enum):
    """base_classes.py:40-42"""
    RK4 = 1
    Euler = 2


def integrator_code(integrator):
    """Accepts this enum, the reference's own enum (same names), or 1/2.
    An unknown integrator is the reference's SyntaxError (base_classes.py:396)."""
    name = getattr(integrator, 'name', None)
    if name == 'RK4' or integrator == 1:
        return 1
    if name == 'Euler' or integrator == 2:
        return 2
    raise SyntaxError('Invalid Integrator Specified. Provided %s. Please choose RK4 or Euler'
                      % (name if name is not None else integrator))
