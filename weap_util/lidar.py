"""`from weap_util.lidar import lidar_to_bitmap` (examples/lidar_example.py:10) -> the HIP rasteriser."""
from red_gym_amd.lidar import lidar_to_bitmap, LidarBitmap, scan_occupancy  # noqa: F401
