from red_gym_amd.f110_env import F110Env  # noqa: F401
