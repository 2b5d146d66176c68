"""Reference-compatible `Simulator` (base_classes.py:451-630) on top of the MI355X BatchSim.

Same constructor arguments, attributes (`agent_poses`, `collisions`, `collision_idx`,
`agents[i].state`) and `set_map / update_params / reset / step` semantics as the reference;
the per-agent Python loop is replaced by one device step over all envs.  Extension:
`num_envs=E` steps E independent copies in lockstep (observations gain a leading env axis).
"""
from enum import Enum

import numpy as np

from . import _ffi
from .core import BatchSim


class Integrator(Enum):      # base_classes.py:40-42
    RK4 = 1
    Euler = 2


def _integrator_code(integrator):
    if isinstance(integrator, Integrator):
        return integrator.value
    if integrator in (1, 2):
        return int(integrator)
    name = getattr(integrator, "name", integrator)
    # base_classes.py:398
    raise SyntaxError("Invalid Integrator Specified. Provided %s. Please choose RK4 or Euler" % (name,))


class ScanNoise(object):
    """The A/B form of the scan noise (noise_mode='table'): the reference draws rng.normal(0, std, B)
    per agent per step from a generator that every agent re-seeds with the SAME seed at reset
    (base_classes.py:204), so the noise is one shared (steps-since-reset, beam) table.  It is
    produced here with NumPy's PCG64 + ziggurat and uploaded; the scan kernel adds row
    `step_count`.  The default (noise_mode='device') draws the same stream on the GPU instead
    (BatchSim.set_noise_rng) with flat memory.  This table stops growing at `max_rows` rows (141 MB at 1080
    beams); an episode longer than that re-uses the rows from the start (the kernel indexes them modulo the
    table length), i.e. its noise repeats — use the device mode for such runs."""

    def __init__(self, seed, num_beams, std_dev=0.01, chunk=256, max_rows=16384):
        self.seed, self.B, self.std, self.chunk = seed, int(num_beams), float(std_dev), int(chunk)
        self.max_rows = int(max_rows)
        self.rng = np.random.default_rng(seed=seed)
        self.rows = np.empty((0, self.B))

    def ensure(self, bsim, needed_rows):
        if self.rows.shape[0] >= needed_rows and bsim.noise_rows == self.rows.shape[0]:
            return
        if needed_rows > self.max_rows:
            if self.rows.shape[0] >= self.max_rows and bsim.noise_rows == self.rows.shape[0]:
                return   # the longest episode any agent can be in is unknown to the host: see step()
            needed_rows = self.max_rows
        target = max(self.chunk, self.rows.shape[0])
        while target < needed_rows:
            target *= 2
        target = min(target, max(self.max_rows, self.chunk))
        if target > self.rows.shape[0]:
            # successive normal(size=B) calls == one normal(size=(k,B)) call (row-major fill)
            extra = self.rng.normal(0., self.std, size=(target - self.rows.shape[0], self.B))
            self.rows = np.concatenate([self.rows, extra], axis=0)
        bsim.set_noise_table(self.rows)


class AgentView(object):
    """What user code reads off `sim.agents[i]` in the reference (RaceCar attributes)."""

    def __init__(self, sim, env, idx):
        self._sim, self._env, self._idx = sim, env, idx

    @property
    def state(self):
        return self._sim._state[self._env * self._sim.num_agents + self._idx]

    @property
    def in_collision(self):
        return bool(self._sim._in_collision[self._env * self._sim.num_agents + self._idx])

    @property
    def params(self):
        return self._sim._agent_params[self._idx]


class Simulator(object):
    def __init__(self, params, num_agents, seed, time_step=0.01, ego_idx=0, integrator=Integrator.RK4,
                 lidar_dist=0.0, num_envs=1, num_beams=1080, fov=4.7, scan_noise_std=0.01, device_id=0,
                 map_layout=_ffi.MAP_DEFAULT, scan_block=0, batched=None, noise_mode='device', step_groups=0):
        self.num_agents = num_agents
        self.num_envs = num_envs
        self.seed = seed
        self.time_step = time_step
        self.ego_idx = ego_idx
        self.params = params
        self._agent_params = [params for _ in range(num_agents)]
        self._b = BatchSim(params, num_envs=num_envs, num_agents=num_agents, num_beams=num_beams, fov=fov,
                           time_step=time_step, integrator=_integrator_code(integrator), lidar_dist=lidar_dist,
                           device_id=device_id, map_layout=map_layout, scan_block=scan_block, step_groups=step_groups)
        N = num_envs * num_agents
        # batched=False: the reference's single-env layout (lists of per-agent values); True: a
        # leading env axis on everything, also for num_envs == 1 (F110VecEnv)
        self._batched = bool(num_envs != 1) if batched is None else bool(batched)
        if not self._batched and num_envs != 1:
            raise ValueError("the single-env layout needs num_envs == 1")
        self.agent_poses = np.empty((num_envs, num_agents, 3)) if self._batched else np.empty((num_agents, 3))
        self.collisions = np.zeros((num_envs, num_agents)) if self._batched else np.zeros((num_agents,))
        self.collision_idx = -1 * np.ones_like(self.collisions)
        self._state = np.zeros((N, 7))
        self._in_collision = np.zeros((N,), dtype=np.int32)
        self.agents = [AgentView(self, 0, i) for i in range(num_agents)]
        # laser_models.py:450-452: noise of std_dev 0.01 from default_rng(seed), re-seeded at reset
        self._noise = None
        if noise_mode not in ('device', 'table'):
            raise ValueError("noise_mode must be 'device' or 'table'")
        if scan_noise_std and scan_noise_std > 0:
            if noise_mode == 'device':
                self._b.set_noise_rng(seed, scan_noise_std)
            else:
                self._noise = ScanNoise(seed, num_beams, scan_noise_std)
        self._steps_since_full_reset = 0
        self._hb = None

    @property
    def batch(self):
        """the underlying BatchSim (device views, unit entry points)"""
        return self._b

    def env_agents(self, env):
        return [AgentView(self, env, i) for i in range(self.num_agents)]

    def set_map(self, map_path, map_ext):
        self._b.set_map(map_path, map_ext)

    def update_params(self, params, agent_idx=-1):
        if agent_idx < 0:
            self._b.set_params(params, -1)
            self._agent_params = [params for _ in range(self.num_agents)]
        elif agent_idx >= 0 and agent_idx < self.num_agents:
            self._b.set_params(params, agent_idx)
            self._agent_params[agent_idx] = params
        else:
            raise IndexError('Index given is out of bounds for list of agents.')

    def reset(self, poses, env_mask=None):
        poses = np.asarray(poses, dtype=np.float64)
        E, A = self.num_envs, self.num_agents
        if not self._batched:
            if poses.shape[0] != A:
                raise ValueError('Number of poses for reset does not match number of agents.')
            flat = poses.reshape(A, 3)
        else:
            if poses.shape[:2] != (E, A):
                raise ValueError('Number of poses for reset does not match number of agents.')
            flat = poses.reshape(E * A, 3)
        self._b.reset(flat, env_mask)
        if env_mask is None or np.all(env_mask):
            self._steps_since_full_reset = 0

    def step(self, control_inputs):
        E, A = self.num_envs, self.num_agents
        if not self._b.has_map:
            raise ValueError('Map is not set for scan simulator.')
        actions = np.asarray(control_inputs, dtype=np.float64).reshape(E * A, 2)
        if self._noise is not None:
            self._noise.ensure(self._b, self._steps_since_full_reset + 1)
        # one ABI call per step (f110_step_host): actions up, the step, the observation written into a
        # page-locked block; the reference hands out fresh arrays every step, so they are copied out
        if E * A * self._b.B * 8 <= (32 << 20):
            hb = self._hb
            if hb is None:
                hb = self._hb = self._b.host_block(("scans", "state", "agent_poses", "collisions", "collision_idx", "in_collision"))
            hb.actions[...] = actions
            self._b.step_host_inplace(hb)
            v = hb.views
            scans, s7 = v["scans"].copy(), v["state"].copy()      # s7: [7][N], the block's own layout
            poses, coll, cidx, inc = v["agent_poses"].T.copy(), v["collisions"].copy(), v["collision_idx"].copy(), v["in_collision"].copy()
        else:
            # big batches through this (host-logic) path: the observation straight into fresh arrays — a page-locked staging
            # block of hundreds of MB and a second host copy would cost more than the call it saves
            self._b.step(actions)
            o = self._b.get("scans", "state", "agent_poses", "collisions", "collision_idx", "in_collision")
            scans, s7, poses, coll, cidx, inc = o["scans"], o["state"].T, o["agent_poses"], o["collisions"], o["collision_idx"], o["in_collision"]
        self._steps_since_full_reset += 1
        self._state = s7.T            # [N][7]
        self._in_collision = inc
        if not self._batched:
            self.agent_poses = poses
            self.collisions = coll
            self.collision_idx = cidx
            # base_classes.py:594-610 — python lists of per-agent values (rows of s7: x, y, steer, v, yaw, yaw rate, slip)
            x, y, vx, th, w = s7[0], s7[1], s7[3], s7[4], s7[5]
            if A == 2:      # (the reference's default: spelled out, a list comprehension costs a frame each)
                observations = {'ego_idx': self.ego_idx, 'scans': [scans[0], scans[1]], 'poses_x': [x[0], x[1]], 'poses_y': [y[0], y[1]],
                                'poses_theta': [th[0], th[1]], 'linear_vels_x': [vx[0], vx[1]], 'linear_vels_y': [0., 0.],
                                'ang_vels_z': [w[0], w[1]], 'collisions': coll}
            elif A == 1:
                observations = {'ego_idx': self.ego_idx, 'scans': [scans[0]], 'poses_x': [x[0]], 'poses_y': [y[0]], 'poses_theta': [th[0]],
                                'linear_vels_x': [vx[0]], 'linear_vels_y': [0.], 'ang_vels_z': [w[0]], 'collisions': coll}
            else:
                r = range(A)
                observations = {'ego_idx': self.ego_idx,
                                'scans': [scans[i] for i in r],
                                'poses_x': [x[i] for i in r],
                                'poses_y': [y[i] for i in r],
                                'poses_theta': [th[i] for i in r],
                                'linear_vels_x': [vx[i] for i in r],
                                'linear_vels_y': [0. for _ in r],
                                'ang_vels_z': [w[i] for i in r],
                                'collisions': coll}
        else:
            self.agent_poses = poses.reshape(E, A, 3)
            self.collisions = coll.reshape(E, A)
            self.collision_idx = cidx.reshape(E, A)
            st = self._state.reshape(E, A, 7)
            observations = {'ego_idx': self.ego_idx,
                            'scans': scans.reshape(E, A, -1),
                            'poses_x': st[:, :, 0], 'poses_y': st[:, :, 1], 'poses_theta': st[:, :, 4],
                            'linear_vels_x': st[:, :, 3], 'linear_vels_y': np.zeros((E, A)),
                            'ang_vels_z': st[:, :, 5], 'collisions': self.collisions}
        return observations
