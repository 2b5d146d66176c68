"""Reference-compatible `RaceCar` (base_classes.py:45-449): one vehicle's physics, scan, iTTC and opponent ray-cast with the
reference's constructor, attributes and methods, evaluated on the MI355X through the unit entry points of the C ABI (one launch
per call).  It exists so that code which imports, type-checks, subclasses or drives `RaceCar` directly keeps working; inside
`Simulator` the per-agent records are columns of device arrays (`Simulator.agents[i]` hands out views), and stepping many cars
goes through `BatchSim`, not through a Python loop over these objects.

Kept from the reference: the scan simulator and the beam tables are CLASS attributes built by the first instance
(:64-67, :118-158) — `RaceCar.scan_simulator`, `.cosines`, `.scan_angles`, `.side_distances` — so a second car with other
`width / lf / lr` inherits the first car's iTTC tables, as it does there."""
import warnings

import numpy as np

from .core import beam_tables
from .laser import ScanSimulator2D
from .sim import Integrator, _integrator_code


class RaceCar(object):
    scan_simulator = None
    cosines = None
    scan_angles = None
    side_distances = None

    def __init__(self, params, seed, is_ego=False, time_step=0.01, num_beams=1080, fov=4.7, integrator=Integrator.Euler, lidar_dist=0.0):
        self.params = params
        self.seed = seed
        self.is_ego = is_ego
        self.time_step = time_step
        self.num_beams = num_beams
        self.fov = fov
        self.integrator = integrator
        self.lidar_dist = lidar_dist
        if self.integrator is Integrator.RK4:
            warnings.warn("Chosen integrator is RK4. This is different from previous versions of the gym.")
        self.state = np.zeros((7, ))     # [x, y, steer_angle, vel, yaw_angle, yaw_rate, slip_angle]
        self.opp_poses = None
        self.accel = 0.0
        self.steer_angle_vel = 0.0
        self.steer_buffer = np.empty((0, ))
        self.steer_buffer_size = 2
        self.in_collision = False
        self.ttc_thresh = 0.005
        if RaceCar.scan_simulator is None:
            self.scan_rng = np.random.default_rng(seed=self.seed)
            RaceCar.scan_simulator = ScanSimulator2D(num_beams, fov)
            # base_classes.py:125-158 (beam_tables is the same arithmetic in NumPy; the device copy lives in the scan simulator's handle)
            RaceCar.scan_angles, RaceCar.cosines, RaceCar.side_distances = beam_tables(num_beams, fov, params)
            RaceCar.scan_simulator.batch.set_beam_tables(RaceCar.scan_angles, RaceCar.cosines, RaceCar.side_distances)

    # -- the handle the unit entry points run on (the class-level scan simulator's)
    @property
    def _b(self):
        return RaceCar.scan_simulator.batch

    def update_params(self, params):
        self.params = params

    def set_map(self, map_path, map_ext):
        RaceCar.scan_simulator.set_map(map_path, map_ext)

    def reset(self, pose):
        self.accel = 0.0
        self.steer_angle_vel = 0.0
        self.in_collision = False
        self.state = np.zeros((7, ))
        self.state[0:2] = pose[0:2]
        self.state[4] = pose[2]
        self.steer_buffer = np.empty((0, ))
        self.scan_rng = np.random.default_rng(seed=self.seed)

    def ray_cast_agents(self, scan):
        new_scan = scan
        pose = np.append(self.state[0:2], self.state[4]).reshape(1, 3)
        for opp_pose in self.opp_poses:
            verts = self._b.get_vertices_batch(np.asarray(opp_pose, dtype=np.float64).reshape(1, 3), self.params['length'], self.params['width'])
            out, _ = self._b.raycast_batch(pose, verts, np.asarray(new_scan, dtype=np.float64).reshape(1, -1))
            new_scan[:] = out[0]      # in place, like laser_models.py:318-346
        return new_scan

    def check_ttc(self, current_scan):
        in_collision = bool(self._b.ttc_batch(np.asarray(current_scan, dtype=np.float64).reshape(1, -1), np.array([self.state[3]]), self.ttc_thresh)[0])
        if in_collision:
            self.state[3:] = 0.
            self.accel = 0.0
            self.steer_angle_vel = 0.0
        self.in_collision = in_collision
        return in_collision

    def update_pose(self, raw_steer, vel):
        cnt = self.steer_buffer.shape[0]
        buf = np.zeros((1, 2))
        buf[0, :cnt] = self.steer_buffer          # index 0 = newest, as np.append(raw_steer, buffer) keeps it
        s1, b1, c1, sp = self._b.update_pose_batch(self.state.reshape(1, 7), buf, np.array([cnt], dtype=np.int32), np.array([[raw_steer, vel]]),
                                                   self.params, self.time_step, _integrator_code(self.integrator), self.lidar_dist)
        self.state = s1[0]
        self.steer_buffer = b1[0, :c1[0]].copy()
        return RaceCar.scan_simulator.scan(sp[0], self.scan_rng)

    def update_opp_poses(self, opp_poses):
        self.opp_poses = opp_poses

    def update_scan(self, agent_scans, agent_index):
        current_scan = agent_scans[agent_index]
        self.check_ttc(current_scan)
        agent_scans[agent_index] = self.ray_cast_agents(current_scan)
