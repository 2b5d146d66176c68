"""The reference's example policy (examples/waypoint_follow.py:145-217) on the device.

`PurePursuitPlanner(conf, wb)` takes the same `conf` namespace the example builds from
the example's yaml (wpt_path, wpt_delim, wpt_rowskip, wpt_xind, wpt_yind, wpt_vind) and
`.plan(pose_x, pose_y, pose_theta, lookahead_distance, vgain)` returns the same `(speed, steer)`
pair.  `.plan_batch` plans for many poses in one launch and `.plan_device` plans for every agent of
a `BatchSim` straight from its device-resident state into a device action buffer, so a closed loop
(plan -> step -> plan ...) never touches the host.
"""
import numpy as np

from .core import BatchSim


class PurePursuitPlanner(object):
    def __init__(self, conf, wb, sim=None, device_id=0):
        self.wheelbase = float(wb)
        self.conf = conf
        self.max_reacquire = 20.      # waypoint_follow.py:153
        self.load_waypoints(conf)
        self._own = sim is None
        self._sim = sim if sim is not None else BatchSim(num_envs=1, num_agents=1, num_beams=64, device_id=device_id)
        self._d_wp = None

    def load_waypoints(self, conf):
        """waypoint_follow.py:157-161; keeps the full table in .waypoints like the reference"""
        if isinstance(conf, np.ndarray):
            self.waypoints = np.asarray(conf, dtype=np.float64)
            self._xyv = np.ascontiguousarray(self.waypoints[:, :3])
        else:
            self.waypoints = np.loadtxt(conf.wpt_path, delimiter=conf.wpt_delim, skiprows=conf.wpt_rowskip)
            self._xyv = np.ascontiguousarray(np.stack([self.waypoints[:, conf.wpt_xind], self.waypoints[:, conf.wpt_yind],
                                                       self.waypoints[:, conf.wpt_vind]], axis=1))
        self._d_wp = None

    def render_waypoints(self, e):
        raise NotImplementedError("rendering is out of scope (DESIGN.md section 8)")

    def plan(self, pose_x, pose_y, pose_theta, lookahead_distance, vgain):
        """waypoint_follow.py:203-217 -> (speed, steering_angle)"""
        a = self.plan_batch(np.array([[pose_x, pose_y, pose_theta]]), lookahead_distance, vgain)[0]
        return float(a[1]), float(a[0])

    def plan_batch(self, poses, lookahead_distance, vgain):
        """poses [m][3] -> actions [m][2] = (steer, speed), the layout env.step takes"""
        return self._sim.pure_pursuit_batch(self._xyv, poses, lookahead_distance, vgain, self.wheelbase, self.max_reacquire)

    def plan_device(self, sim, d_actions, lookahead_distance, vgain):
        """actions for all agents of `sim` from its live poses, device to device"""
        if self._d_wp is None or self._d_wp_owner is not sim:
            self._d_wp = sim.device_array(self._xyv.shape)
            self._d_wp.upload(self._xyv)
            self._d_wp_owner = sim
        sim.pure_pursuit_device(self._d_wp, self._xyv.shape[0], d_actions, lookahead_distance, vgain, self.wheelbase,
                                self.max_reacquire)

    def close(self):
        if self._own and self._sim is not None:
            self._sim.close()
        self._sim = None
