from red_gym_amd.f110_env import F110Env  # noqa: F401
from red_gym_amd.base_classes import *  # noqa: F401,F403
from red_gym_amd.laser_models import *  # noqa: F401,F403
