"""`F110Env` — the reference's gym façade (f110_env.py:53-418) over the MI355X simulator, plus
`F110VecEnv`, the batched form RL loops should use (E envs per step, observations as arrays).

Kept from the reference: constructor kwargs and defaults (:104-159), `reset(poses)` that
advances one zero-action step and returns a 4-tuple (:306-349), `step(action)` returning
(obs, reward=timestep, done, info={'checkpoint_done': ...}) (:263-304), the lap/finish logic
(:204-246), `update_map`, `update_params`, `add_render_callback`.  Rendering (pyglet) is out of
scope for this build: `render()` raises NotImplementedError.
`gym` is optional: when importable F110Env subclasses gym.Env, otherwise `object`.
"""
import os

import numpy as np

from . import _ffi
from .core import DEFAULT_PARAMS
from .sim import Integrator, Simulator

try:  # pragma: no cover - gym is absent from the build image
    import gym as _gym
    _EnvBase = _gym.Env
except Exception:  # noqa: BLE001
    _EnvBase = object

_PKG_MAPS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "maps")


def _resolve_map_path(kwargs):
    """f110_env.py:108-120: named maps ship with the package, anything else is a path stem."""
    if 'map' not in kwargs:
        return 'vegas', os.path.join(_PKG_MAPS, 'vegas.yaml')
    name = kwargs['map']
    if name in ('berlin', 'skirk', 'levine', 'vegas', 'stata_basement'):
        return name, os.path.join(_PKG_MAPS, name + '.yaml')
    return name, name + '.yaml'


class _LapLogic(object):
    """The start/finish bookkeeping of F110Env._check_done (f110_env.py:204-246), vectorised
    over a leading env axis: arrays are [E][A]."""

    def __init__(self, num_envs, num_agents, ego_idx):
        self.E, self.A, self.ego = num_envs, num_agents, ego_idx
        self.start_xs = np.zeros((num_envs, num_agents))
        self.start_ys = np.zeros((num_envs, num_agents))
        self.start_thetas = np.zeros((num_envs, num_agents))
        self.rot_c = np.ones((num_envs,))
        self.rot_s = np.zeros((num_envs,))
        self.near_starts = np.ones((num_envs, num_agents), dtype=bool)
        self.toggle_list = np.zeros((num_envs, num_agents))
        self.lap_times = np.zeros((num_envs, num_agents))
        self.lap_counts = np.zeros((num_envs, num_agents))
        self.current_time = np.zeros((num_envs,))
        self._c0, self._s0, self._sx0, self._sy0 = 1.0, 0.0, [0.0] * num_agents, [0.0] * num_agents

    def reset(self, poses, env_mask=None):
        m = np.ones((self.E,), dtype=bool) if env_mask is None else np.asarray(env_mask, dtype=bool)
        poses = np.asarray(poses, dtype=np.float64).reshape(self.E, self.A, 3)
        self.current_time[m] = 0.0
        self.near_starts[m] = True
        self.toggle_list[m] = 0
        self.start_xs[m] = poses[m, :, 0]
        self.start_ys[m] = poses[m, :, 1]
        self.start_thetas[m] = poses[m, :, 2]
        th = -self.start_thetas[:, self.ego]
        self.rot_c[m] = np.cos(th)[m]
        self.rot_s[m] = np.sin(th)[m]
        # update_single's constants as plain floats (env 0): the single-env step counts microseconds
        self._c0, self._s0 = float(self.rot_c[0]), float(self.rot_s[0])
        self._sx0, self._sy0 = [float(v) for v in self.start_xs[0]], [float(v) for v in self.start_ys[0]]

    def update_single(self, poses_x, poses_y, collisions, timestep):
        """update() for ONE env in plain Python floats: the same IEEE operations in the same order, without ~25 NumPy
        calls on arrays of A elements (the single-env F110Env.step is latency-bound, every microsecond of host time
        counts).  returns done (bool), checkpoint_done [A]"""
        ct = float(self.current_time[0]) + timestep
        self.current_time[0] = ct
        c, s = self._c0, self._s0
        sx, sy = self._sx0, self._sy0
        near, tog = self.near_starts[0], self.toggle_list[0]
        all4 = True
        for i in range(self.A):
            px = float(poses_x[i]) - sx[i]
            py = float(poses_y[i]) - sy[i]
            dx = c * px + (-s) * py
            ty = s * px + c * py
            if ty > 2:
                ty = ty - 2
            elif ty < -2:
                ty = -2 - ty
            else:
                ty = 0.0
            closes = dx ** 2 + ty ** 2 <= 0.1
            if closes != bool(near[i]):
                near[i] = closes
                tog[i] += 1
            t = float(tog[i])
            self.lap_counts[0, i] = t // 2
            if t < 4:
                self.lap_times[0, i] = ct
                all4 = False
        done = bool(collisions[self.ego] != 0) or all4
        return done, tog >= 4

    def update(self, poses_x, poses_y, collisions, timestep):
        """returns done[E], checkpoint_done[E][A]"""
        left_t, right_t = 2, 2
        self.current_time = self.current_time + timestep
        px = np.asarray(poses_x, dtype=np.float64).reshape(self.E, self.A) - self.start_xs
        py = np.asarray(poses_y, dtype=np.float64).reshape(self.E, self.A) - self.start_ys
        c, s = self.rot_c[:, None], self.rot_s[:, None]
        dx = c * px + (-s) * py          # start_rot @ [px; py], f110_env.py:223,331
        temp_y = s * px + c * py
        idx1 = temp_y > left_t
        idx2 = temp_y < -right_t
        temp_y = np.where(idx1, temp_y - left_t, np.where(idx2, -right_t - temp_y, 0.0))
        dist2 = dx ** 2 + temp_y ** 2
        closes = dist2 <= 0.1
        entered = closes & ~self.near_starts
        left = ~closes & self.near_starts
        self.near_starts = np.where(entered, True, np.where(left, False, self.near_starts))
        self.toggle_list = self.toggle_list + (entered | left)
        self.lap_counts[...] = self.toggle_list // 2
        running = self.toggle_list < 4
        self.lap_times[...] = np.where(running, self.current_time[:, None], self.lap_times)
        col = np.asarray(collisions).reshape(self.E, self.A)
        done = (col[:, self.ego] != 0) | np.all(self.toggle_list >= 4, axis=1)
        return done, self.toggle_list >= 4


class F110Env(_EnvBase):
    metadata = {'render.modes': ['human', 'human_fast']}
    render_callbacks = []

    def __init__(self, **kwargs):
        self.seed = kwargs.get('seed', 12345)
        self.map_name, self.map_path = _resolve_map_path(kwargs)
        self.map_ext = kwargs.get('map_ext', '.png')
        self.params = kwargs.get('params', dict(DEFAULT_PARAMS))
        self.num_agents = kwargs.get('num_agents', 2)
        self.timestep = kwargs.get('timestep', 0.01)
        self.ego_idx = kwargs.get('ego_idx', 0)
        self.integrator = kwargs.get('integrator', Integrator.RK4)
        self.lidar_dist = kwargs.get('lidar_dist', 0.0)
        self.start_thresh = 0.5
        self.poses_x, self.poses_y, self.poses_theta = [], [], []
        self.collisions = np.zeros((self.num_agents,))
        self._lap = _LapLogic(1, self.num_agents, self.ego_idx)
        # f110_env.py:192 does NOT hand ego_idx to its Simulator: obs['ego_idx'] is 0 whatever the env's
        # ego_idx is (pinned by tests/golden/env_episode_2agents.npz); ego_idx only steers _check_done
        self.sim = Simulator(self.params, self.num_agents, self.seed, time_step=self.timestep,
                             integrator=self.integrator, lidar_dist=self.lidar_dist,
                             device_id=kwargs.get('device_id', 0),
                             map_layout=kwargs.get('map_layout', _ffi.MAP_DEFAULT))
        self.sim.set_map(self.map_path, self.map_ext)
        self.render_obs = None
        self.current_obs = None

    # attributes user code reads off the reference env
    lap_times = property(lambda self: self._lap.lap_times[0])
    lap_counts = property(lambda self: self._lap.lap_counts[0])
    current_time = property(lambda self: float(self._lap.current_time[0]))
    toggle_list = property(lambda self: self._lap.toggle_list[0])
    near_starts = property(lambda self: self._lap.near_starts[0])
    start_xs = property(lambda self: self._lap.start_xs[0])
    start_ys = property(lambda self: self._lap.start_ys[0])
    start_thetas = property(lambda self: self._lap.start_thetas[0])

    def step(self, action):
        obs = self.sim.step(action)
        obs['lap_times'] = self._lap.lap_times[0]
        obs['lap_counts'] = self._lap.lap_counts[0]
        self.current_obs = obs
        self.render_obs = {k: obs[k] for k in ('ego_idx', 'poses_x', 'poses_y', 'poses_theta', 'lap_times', 'lap_counts')}
        reward = self.timestep
        self.poses_x, self.poses_y, self.poses_theta = obs['poses_x'], obs['poses_y'], obs['poses_theta']
        self.collisions = obs['collisions']
        done, toggles = self._lap.update_single(obs['poses_x'], obs['poses_y'], obs['collisions'], self.timestep)
        info = {'checkpoint_done': toggles}
        return obs, reward, done, info

    def reset(self, poses):
        poses = np.asarray(poses, dtype=np.float64)
        self.collisions = np.zeros((self.num_agents,))
        self.sim.reset(poses)               # raises ValueError on a pose-count mismatch
        self._lap.reset(poses)
        action = np.zeros((self.num_agents, 2))
        return self.step(action)            # f110_env.py:337-338: reset advances one step

    def update_map(self, map_path, map_ext):
        self.sim.set_map(map_path, map_ext)

    def update_params(self, params, index=-1):
        self.sim.update_params(params, agent_idx=index)

    def add_render_callback(self, callback_func):
        F110Env.render_callbacks.append(callback_func)

    def render(self, mode='human'):
        assert mode in ['human', 'human_fast']
        raise NotImplementedError("rendering (pyglet) is outside this build's scope; use obs['poses_*']")


class F110VecEnv(object):
    """E independent F110 environments stepped by one device launch sequence.

    reset(poses[E][A][3], env_mask=None) / step(actions[E][A][2]) -> (obs, reward, done[E], info)
    with array observations (leading env axis).

    Resets.  reset(poses) of every env is the reference's reset(): re-seat + one zero-action step,
    whose observation is returned (f110_env.py:337-338).  A PARTIAL reset — reset(poses, env_mask)
    or `auto_reset=True`, which re-seats finished envs at their start poses inside step() (mask
    reset in place) — only re-seats: nobody is stepped, the envs that are in the middle
    of an episode are not disturbed, and a re-seated env's first observation arrives with the next
    step().  reset(poses, env_mask) therefore returns the previous step's tuple with `done`
    cleared for the re-seated envs.

    device_logic=True runs the lap / done bookkeeping (F110Env._check_done) and the auto-reset on
    the GPU: a step is ONE ABI call (f110_step_host) — the step's kernels read the actions in place
    from a page-locked buffer (`mapped_actions`; False: a staging copy first) and one kernel writes
    `done`, the lap arrays and the requested observation columns straight into page-locked host
    memory (scans: one DMA copy).  The arrays step() returns in that mode are VIEWS of that block,
    the same objects every step, overwritten by the next step(): copy what you keep, or pass
    copy_obs=True (the block stays valid as long as any array views it, also after close()).
    obs_fields selects the fields put into `obs` ('scans' is 8.6 KB per agent); episode_fields
    the episode columns brought back next to `done` (default: lap_times, lap_counts in obs and
    toggle_list, near_starts, checkpoint_done in info; () for the leanest loop); everything stays
    available in HBM through `device_views()`.  `env.action_buffer` ([E][A][2], page-locked) can be
    filled in place and step(None) called: no copy of the actions at all.  step_async() /
    step_wait() split the call the way gym.vector.VectorEnv does (between the two, `action_buffer` belongs to the
    GPU: the kernels read it in place — write the next actions only after step_wait()).

    Domain randomisation over tracks: `extra_maps=[(yaml_path, ext), ...]` registers further maps
    (slots 1, 2, ...; `map` is slot 0) and `env_map=[slot per env]` assigns them; `set_env_maps()`
    re-assigns later.
    """

    # every key of the reference's observation (base_classes.py:594-610, docs/api/obv.rst:6-14)
    _ALL = ("scans", "poses_x", "poses_y", "poses_theta", "linear_vels_x", "linear_vels_y", "ang_vels_z", "collisions")

    # what the device episode logic can bring back next to `done` (info keys + the two lap arrays of obs)
    _EPISODE = ("lap_times", "lap_counts", "toggle_list", "near_starts", "checkpoint_done")

    def __init__(self, num_envs, auto_reset=False, device_logic=False, obs_fields=None, copy_obs=False,
                 episode_fields=None, mapped_actions=True, spin_wait=False, fuse_host_block=True, poll_wait=True, **kwargs):
        self.num_envs = int(num_envs)
        self.seed = kwargs.get('seed', 12345)
        self.map_name, self.map_path = _resolve_map_path(kwargs)
        self.map_ext = kwargs.get('map_ext', '.png')
        self.params = kwargs.get('params', dict(DEFAULT_PARAMS))
        self.num_agents = kwargs.get('num_agents', 2)
        self.timestep = kwargs.get('timestep', 0.01)
        self.ego_idx = kwargs.get('ego_idx', 0)
        self.auto_reset = auto_reset
        self.device_logic = bool(device_logic)
        self.obs_fields = tuple(self._ALL if obs_fields is None else obs_fields)
        self._lap = _LapLogic(self.num_envs, self.num_agents, self.ego_idx)
        self.sim = Simulator(self.params, self.num_agents, self.seed, time_step=self.timestep,
                             integrator=kwargs.get('integrator', Integrator.RK4),   # (no ego_idx: see obs['ego_idx'] below)
                             lidar_dist=kwargs.get('lidar_dist', 0.0), num_envs=self.num_envs,
                             num_beams=kwargs.get('num_beams', 1080), fov=kwargs.get('fov', 4.7),
                             scan_noise_std=kwargs.get('scan_noise_std', 0.01),
                             device_id=kwargs.get('device_id', 0),
                             map_layout=kwargs.get('map_layout', _ffi.MAP_DEFAULT), batched=True,
                             noise_mode=kwargs.get('noise_mode', 'device'), step_groups=kwargs.get('step_groups', 0))
        self.sim.set_map(self.map_path, self.map_ext)
        self._last = None
        self.map_slots = [(self.map_path, self.map_ext)]
        for path, ext in kwargs.get('extra_maps', ()):
            self.sim.batch.add_map(path, ext)
            self.map_slots.append((path, ext))
        if kwargs.get('env_map') is not None:
            self.set_env_maps(kwargs['env_map'])
        self._start_poses = None
        self._d_actions = None
        self.copy_obs = bool(copy_obs)
        self.episode_fields = tuple(self._EPISODE if episode_fields is None else episode_fields)
        self.mapped_actions = bool(mapped_actions)
        self.spin_wait = bool(spin_wait)
        self.fuse_host_block = bool(fuse_host_block)
        self.poll_wait = bool(poll_wait)
        if self.device_logic:
            b = self.sim.batch
            b.episode_init(self.ego_idx)
            self._d_actions = b.device_array((self.num_envs * self.num_agents, 2))
            self._build_host_block()

    def _build_host_block(self):
        """the page-locked block f110_step_host fills, and the (obs, reward, done, info) tuple of views into it
        that step() hands out — built once: a step is one ABI call, no per-step allocation"""
        E, A, b = self.num_envs, self.num_agents, self.sim.batch
        want = ["done"]
        st_fields = {"poses_x": 0, "poses_y": 1, "poses_theta": 4, "linear_vels_x": 3, "ang_vels_z": 5}
        if any(f in st_fields for f in self.obs_fields):
            want.append("state")
        want += [f for f in ("collisions", "scans") if f in self.obs_fields]
        ep_names = {"lap_times": "lap_times", "lap_counts": "lap_counts", "toggle_list": "toggles",
                    "near_starts": "near_starts", "checkpoint_done": "checkpoint_done"}
        for f in self.episode_fields:
            want.append(ep_names[f])      # KeyError: not an episode field
        hb = self._hb = b.host_block(want)
        v = hb.views
        self.action_buffer = hb.actions.reshape(E, A, 2)   # write actions here and call step(None): no copy at all
        # obs['ego_idx'] is 0 whatever the env's ego_idx is: the reference env never hands ego_idx to its Simulator
        # (f110_env.py:192) — the same quirk F110Env reproduces; ego_idx steers the done rule only
        obs = {'ego_idx': 0}
        for f in self.obs_fields:
            if f in st_fields:
                obs[f] = v["state"][st_fields[f]].reshape(E, A)
            elif f == "linear_vels_y":
                obs[f] = np.zeros((E, A))      # base_classes.py:603: always 0. in the reference
            elif f == "scans":
                obs[f] = v["scans"].reshape(E, A, -1)
            else:
                obs[f] = v[f].reshape(E, A)
        info = {}
        for f in self.episode_fields:
            arr = v[ep_names[f]].reshape(E, A)
            if f in ("near_starts", "checkpoint_done"):
                arr = arr.view(np.bool_)
            (obs if f in ("lap_times", "lap_counts") else info)[f] = arr
        self._ret_views = (obs, self.timestep, v["done"].view(np.bool_), info)

    def update_params_batch(self, params):
        """a vehicle parameter set per agent of every env ([E*A] dicts or [E*A][18] array; None: back
        to the per-slot sets of update_params)"""
        self.sim.batch.set_params_batch(params)

    def set_env_maps(self, env_map):
        """env_map [num_envs]: which registered track each env runs on (None: all on slot 0)"""
        self.sim.batch.set_env_maps(env_map)
        self.env_map = None if env_map is None else np.asarray(env_map, dtype=np.int32).copy()

    def device_views(self):
        v = self.sim.batch.device_views()
        if self.device_logic:
            v.update(self.sim.batch.episode_device_views())
        return v

    def reset(self, poses, env_mask=None):
        poses = np.asarray(poses, dtype=np.float64).reshape(self.num_envs, self.num_agents, 3)
        self._start_poses = poses.copy() if self._start_poses is None or env_mask is None else \
            np.where(np.asarray(env_mask, dtype=bool)[:, None, None], poses, self._start_poses)
        if self.device_logic:
            self.sim.batch.episode_reset(poses.reshape(-1, 3), env_mask)
            if env_mask is None or np.all(env_mask):
                self.sim._steps_since_full_reset = 0
        else:
            self.sim.reset(poses, env_mask)
            self._lap.reset(poses, env_mask)
        if env_mask is not None and not np.all(env_mask) and self._last is not None:
            # partial reset: re-seat only (class docstring); envs in mid-episode are not stepped
            obs, reward, done, info = self._last
            done = np.where(np.asarray(env_mask, dtype=bool), False, done)
            self._last = (obs, reward, done, info)
            return self._last
        return self.step(np.zeros((self.num_envs, self.num_agents, 2)))

    def _step_device(self, actions, sync=True):
        b, hb = self.sim.batch, self._hb
        if self.sim._noise is not None:
            self.sim._noise.ensure(b, self.sim._steps_since_full_reset + 1)
        if actions is not None and actions is not self.action_buffer:
            hb.actions[...] = np.asarray(actions, dtype=np.float64).reshape(hb.actions.shape)
        b.step_host(hb, None, auto_reset=self.auto_reset, sync=sync, mapped_actions=self.mapped_actions, spin=self.spin_wait, fuse=self.fuse_host_block, poll=self.poll_wait)
        self.sim._steps_since_full_reset += 1
        if not sync:
            return None
        return self._collect()

    def _collect(self):
        obs, r, done, info = self._ret_views
        if self.copy_obs:
            obs = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in obs.items()}
            done, info = done.copy(), {k: v.copy() for k, v in info.items()}
        self._last = (obs, r, done, info)
        return self._last

    # gym.vector.VectorEnv's split: step_async() enqueues the whole step and returns at once, step_wait()
    # blocks until the observation block is complete (the host may do other work in between)
    def step_async(self, actions):
        if not self.device_logic:
            raise ValueError("step_async needs device_logic=True")
        self._step_device(actions, sync=False)

    def step_wait(self):
        self.sim.batch.sync()
        return self._collect()

    def step(self, actions):
        if self.device_logic:
            return self._step_device(actions)
        if actions is None:
            raise ValueError("step(None) (actions taken from env.action_buffer) needs device_logic=True")
        obs = self.sim.step(actions)
        done, toggles = self._lap.update(obs['poses_x'], obs['poses_y'], obs['collisions'], self.timestep)
        obs['lap_times'] = self._lap.lap_times
        obs['lap_counts'] = self._lap.lap_counts
        info = {'checkpoint_done': toggles, 'toggle_list': self._lap.toggle_list.copy(),
                'near_starts': self._lap.near_starts.copy()}
        if self.auto_reset and done.any():
            self.sim.reset(self._start_poses, done)
            self._lap.reset(self._start_poses, done)
        self._last = (obs, self.timestep, done, info)
        return self._last
