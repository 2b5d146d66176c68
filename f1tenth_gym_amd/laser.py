"""Reference-compatible `ScanSimulator2D` (laser_models.py:348-457) backed by the MI355X
ray-march kernel.  `scan(pose, rng)` keeps the reference's contract: a noise-free device scan,
then `rng.normal(0, std_dev, num_beams)` added on the host exactly as laser_models.py:450-452.
`scan_batch(poses)` is the batched extension (M poses in one launch)."""
import numpy as np

from . import _ffi
from .core import BatchSim, load_map_files


class ScanSimulator2D(object):
    def __init__(self, num_beams, fov, eps=0.0001, theta_dis=2000, max_range=30.0, device_id=0,
                 map_layout=_ffi.MAP_DEFAULT):
        self.num_beams = num_beams
        self.fov = fov
        self.eps = eps
        self.theta_dis = theta_dis
        self.max_range = max_range
        self.angle_increment = self.fov / (self.num_beams - 1)
        self.theta_index_increment = theta_dis * self.angle_increment / (2. * np.pi)
        self.orig_c = None
        self.orig_s = None
        self.orig_x = None
        self.orig_y = None
        self.map_height = None
        self.map_width = None
        self.map_resolution = None
        self._dt = None
        theta_arr = np.linspace(0.0, 2 * np.pi, num=theta_dis)
        self.sines = np.sin(theta_arr)
        self.cosines = np.cos(theta_arr)
        self._b = BatchSim(None, num_envs=1, num_agents=1, num_beams=num_beams, fov=fov, eps=eps,
                           theta_dis=theta_dis, max_range=max_range, device_id=device_id, map_layout=map_layout)

    @property
    def batch(self):
        return self._b

    @property
    def dt(self):
        """the distance table (read back from HBM on first use)"""
        if self._dt is None and self.map_height is not None:
            self._dt = self._b.get_map_dt()
        return self._dt

    def set_map(self, map_path, map_ext):
        img, res, origin = load_map_files(map_path, map_ext)
        self._b.set_map_image(img, res, origin)
        self.map_height, self.map_width = img.shape
        self.map_resolution = res
        self.origin = origin
        self.orig_x, self.orig_y = origin[0], origin[1]
        self.orig_s, self.orig_c = np.sin(origin[2]), np.cos(origin[2])
        self._dt = None
        return True

    def set_map_dt(self, dt, resolution, origin):
        self._b.set_map_dt(dt, resolution, origin)
        self.map_height, self.map_width = np.asarray(dt).shape
        self.map_resolution = resolution
        self.origin = list(origin)
        self.orig_x, self.orig_y = origin[0], origin[1]
        self.orig_s, self.orig_c = np.sin(origin[2]), np.cos(origin[2])
        self._dt = None
        return True

    def scan(self, pose, rng, std_dev=0.01):
        if self.map_height is None:
            raise ValueError('Map is not set for scan simulator.')
        scan = self._b.scan_batch(np.asarray(pose, dtype=np.float64).reshape(1, 3))[0]
        if rng is not None:
            noise = rng.normal(0., std_dev, size=self.num_beams)
            scan += noise
        return scan

    def scan_batch(self, poses, want_hits=False, want_lookups=False):
        if self.map_height is None:
            raise ValueError('Map is not set for scan simulator.')
        return self._b.scan_batch(poses, want_hits=want_hits, want_lookups=want_lookups)

    def get_increment(self):
        return self.angle_increment
