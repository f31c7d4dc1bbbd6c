"""Map / raceline loaders with the reference's semantics.

load_map follows ScanSimulator2D.set_map (laser_models.py:383-427): image path =
yaml path with its extension replaced by map_ext (:398), PIL open +
FLIP_TOP_BOTTOM (:399), <=128 -> occupied, >128 -> free (:403-404), only
`resolution` and `origin[0:3]` of the YAML are used (:413-422).
"""
import os

import numpy as np

ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'assets')
BUILTIN_MAPS = ('berlin', 'skirk', 'levine')  # names f110_env.py:109-114 resolves to packaged maps
DEFAULT_MAP = 'vegas'  # packaged map used only when the `map` keyword is ABSENT (f110_env.py:117-118)


class MapData(object):
    __slots__ = ('free', 'height', 'width', 'resolution', 'orig_x', 'orig_y', 'orig_c', 'orig_s', 'yaml_path')


def builtin_map_yaml(name):
    return os.path.join(ASSETS, 'maps', name + '.yaml')


def load_map(map_path, map_ext):
    import yaml
    from PIL import Image
    img_path = os.path.splitext(map_path)[0] + map_ext
    if not os.path.exists(img_path):
        # the reference fails inside PIL here (laser_models.py:399); e.g. the packaged 'levine' ships a YAML
        # without an image, there as here
        raise FileNotFoundError('map image %s not found (yaml %s, map_ext %r)' % (img_path, map_path, map_ext))
    img = np.array(Image.open(img_path).transpose(Image.FLIP_TOP_BOTTOM)).astype(np.float64)
    if img.ndim != 2:
        raise ValueError('map image must be single-channel, got shape %s' % (img.shape,))
    with open(map_path, 'r') as f:
        meta = yaml.safe_load(f)
    m = MapData()
    m.free = np.ascontiguousarray((img > 128.).astype(np.uint8))
    m.height, m.width = img.shape
    m.resolution = float(meta['resolution'])
    origin = meta['origin']
    m.orig_x, m.orig_y = float(origin[0]), float(origin[1])
    m.orig_s, m.orig_c = float(np.sin(origin[2])), float(np.cos(origin[2]))
    m.yaml_path = map_path
    return m


def load_waypoints(path, delimiter=';', skiprows=3):
    """examples/waypoint_follow.py:162 raceline CSV (3 header rows, ';')."""
    return np.loadtxt(path, delimiter=delimiter, skiprows=skiprows)
