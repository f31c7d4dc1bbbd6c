from red_gym_amd.laser_models import ScanSimulator2D  # noqa: F401
