from red_gym_amd.base_classes import Integrator, Simulator, RaceCar  # noqa: F401
