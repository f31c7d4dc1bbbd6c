"""ScanSimulator2D with the reference's interface (gym/f110_gym/envs/laser_models.py:348-457),
served by the HIP scan kernel (f110_scan through the C ABI)."""
import numpy as np

__all__ = ['ScanSimulator2D']


class ScanSimulator2D(object):
    def __init__(self, num_beams, fov, eps=0.0001, theta_dis=2000, max_range=30.0, device=0):
        from .engine import Engine
        self.num_beams, self.fov, self.eps = num_beams, fov, eps
        self.theta_dis, self.max_range = theta_dis, max_range
        self.angle_increment = self.fov / (self.num_beams - 1)
        self.theta_index_increment = theta_dis * self.angle_increment / (2. * np.pi)
        self.map_height = self.map_width = self.map_resolution = None
        self._eng = Engine(num_envs=1, num_agents=1, fov=fov, num_beams=num_beams, eps=eps, theta_dis=theta_dis,
                           max_range=max_range, device=device, noise_std=0)
        self.sines, self.cosines = self._eng.sines, self._eng.cosines

    def set_map(self, map_path, map_ext):
        self._eng.set_map(map_path, map_ext)
        m = self._eng.map
        self.map_height, self.map_width, self.map_resolution = m.height, m.width, m.resolution
        self.orig_x, self.orig_y, self.orig_c, self.orig_s = m.orig_x, m.orig_y, m.orig_c, m.orig_s
        return True

    @property
    def dt(self):
        return self._eng.get_map_dt()

    def scan(self, pose, rng, std_dev=0.01):
        if self.map_height is None:
            raise ValueError('Map is not set for scan simulator.')
        scan = self._eng.scan(np.asarray(pose, dtype=np.float64).reshape(1, 3))[0].cpu().numpy()
        if rng is not None:
            noise = rng.normal(0., std_dev, size=self.num_beams)
            scan += noise
        return scan

    def scan_batch(self, poses):
        """Extension: [n,3] poses -> [n,num_beams] torch tensor on the device (noise off)."""
        if self.map_height is None:
            raise ValueError('Map is not set for scan simulator.')
        return self._eng.scan(poses)

    def get_increment(self):
        return self.angle_increment
