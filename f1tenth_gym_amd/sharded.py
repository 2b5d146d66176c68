"""`ShardedVecEnv` — E independent environments over several MI355X from ONE process (SURVEY §8e "one process
with 8 handles"): contiguous env blocks, one `F110VecEnv` handle per entry of `devices`, one worker thread per
handle, no collective on the step path.  What the reference runs as a serial agent loop inside one Simulator
(base_classes.py:568-585) is already a batch per handle; this class is the layer above it that a user of 262 144
envs on 8 GPUs would otherwise have to copy out of bench.py.

    env = ShardedVecEnv(262144 // 2, devices=range(8), map=..., map_ext='.png', num_agents=2, auto_reset=True)
    obs, reward, done, info = env.reset(poses)          # poses [E][A][3]
    obs, reward, done, info = env.step(actions)         # actions [E][A][2]; arrays carry the global env axis

Results are bit-equal to ONE handle stepping the same envs: environments never interact, every agent's noise stream
restarts from the same seed at its reset (base_classes.py:204), and each shard runs the very kernels a single handle
runs over that env range (tests/test_gpu_round6.py compares shardings of one device against a single handle).
A device id may appear more than once in `devices` (several handles on one GPU — how a 1-GPU box tests this class).

The optional observation gather (`gather_obs=True`, BASELINE configs[3]) leaves EVERY shard's scans + 7 scalars per
agent on EVERY device after each step — one RCCL communicator over the handles, `f110_comm_all_gather_obs` on each
shard's stream (float32 transport / gather to one root: `gather_f32=True`, `gather_root=k`); `gathered_views()`
hands out the per-device receive buffers ([shards][N_k][B] and [shards][7][N_k]).  It needs equal shards.

For a device-resident RL loop use `shards[k].device_views()` (actions and observations never leave the GPU that owns
them); the array-returning step() here is the host-driven form, like F110VecEnv's.
"""
import queue
import threading

import numpy as np

from .env import F110VecEnv


class _Worker(threading.Thread):
    """one shard's calls, in order, on a thread of its own (ctypes releases the GIL: the shards' steps overlap)"""

    def __init__(self, idx):
        threading.Thread.__init__(self, name="f110-shard-%d" % idx, daemon=True)
        self.q = queue.Queue()
        self.start()

    def run(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            fn, box, ev = item
            try:
                box["v"] = fn()
            except BaseException as ex:  # noqa: BLE001 - re-raised in the caller's thread
                box["e"] = ex
            ev.set()

    def submit(self, fn):
        box, ev = {}, threading.Event()
        self.q.put((fn, box, ev))
        return box, ev

    def stop(self):
        self.q.put(None)


def _wait(pending):
    out, err = [], None
    for box, ev in pending:
        ev.wait()
        err = err or box.get("e")
        out.append(box.get("v"))
    if err is not None:
        raise err
    return out


class ShardedVecEnv(object):
    def __init__(self, num_envs, devices=(0,), gather_obs=False, gather_f32=False, gather_root=None, shard_sizes=None, threaded_step=None, **kwargs):
        self.num_envs = E = int(num_envs)
        self.devices = [int(d) for d in devices]
        K = len(self.devices)
        if K < 1:
            raise ValueError("devices must name at least one HIP device")
        if shard_sizes is None:
            base, rem = divmod(E, K)
            shard_sizes = [base + (1 if k < rem else 0) for k in range(K)]   # contiguous blocks, sizes differ by at most one
        self.shard_sizes = [int(n) for n in shard_sizes]
        if len(self.shard_sizes) != K or sum(self.shard_sizes) != E or min(self.shard_sizes) < 1:
            raise ValueError("shard_sizes must give every device at least one env and add up to num_envs")
        self.bounds = np.concatenate([[0], np.cumsum(self.shard_sizes)]).astype(np.int64)
        self.num_agents = kwargs.get('num_agents', 2)
        self.timestep = kwargs.get('timestep', 0.01)
        kwargs.setdefault('device_logic', True)
        env_map = kwargs.pop('env_map', None)
        self._workers = [_Worker(k) for k in range(K)]
        self.shards = [None] * K

        def make(k):
            kw = dict(kwargs, device_id=self.devices[k])
            if env_map is not None:
                kw['env_map'] = np.asarray(env_map)[self.bounds[k]:self.bounds[k + 1]]
            self.shards[k] = F110VecEnv(self.shard_sizes[k], **kw)
        try:
            _wait([w.submit(lambda k=k: make(k)) for k, w in enumerate(self._workers)])
        except BaseException:
            self.close()
            raise
        self.device_logic = self.shards[0].device_logic
        # How step() reaches the shards.  With the episode logic on the devices a shard's step splits into an enqueue that returns at
        # once and a wait (F110VecEnv.step_async / step_wait): the caller's thread enqueues every shard, then waits for every shard —
        # the GPUs run side by side and no thread hand-off (two futex wake-ups per shard and step) is paid.  The worker threads carry
        # the step when it cannot be split (host-side episode logic) or when the observation includes the scans (each worker copies its
        # own block out of its page-locked memory: those copies are what should overlap).  threaded_step=True / False forces one form.
        if threaded_step is None:
            threaded_step = not self.device_logic or "scans" in self.shards[0].obs_fields
        self.threaded_step = bool(threaded_step) or not self.device_logic
        self._out = None
        self._gather = None
        if gather_obs:
            self._init_gather(bool(gather_f32), gather_root)

    # ------------------------------------------------------------------ plumbing
    def _slices(self, arr):
        return [arr[self.bounds[k]:self.bounds[k + 1]] for k in range(len(self.shards))]

    def _each(self, fn):
        """fn(k, shard) on every shard's worker thread; -> list of results"""
        return _wait([w.submit(lambda k=k: fn(k, self.shards[k])) for k, w in enumerate(self._workers)])

    def _assemble(self, parts):
        """parts[k] = (obs, reward, done, info) of shard k -> one tuple over the global env axis.  The output arrays are
        allocated once and refilled (like F110VecEnv's views they are OVERWRITTEN by the next step: copy what you keep)."""
        if len(parts) == 1:
            return parts[0]           # one shard: its own tuple (views of its page-locked block, as F110VecEnv hands them out)
        if self._out is None:
            def alloc(v):
                return np.empty((self.num_envs,) + v.shape[1:], dtype=v.dtype)
            o0, _, d0, i0 = parts[0]
            self._out = ({k: (alloc(v) if isinstance(v, np.ndarray) else v) for k, v in o0.items()}, alloc(np.asarray(d0)),
                         {k: alloc(v) for k, v in i0.items()})
        obs, done, info = self._out

        def fill(k, _shard=None):
            o, _, d, i = parts[k]
            lo, hi = self.bounds[k], self.bounds[k + 1]
            for name, v in o.items():
                if isinstance(v, np.ndarray):
                    obs[name][lo:hi] = v
            done[lo:hi] = d
            for name, v in i.items():
                info[name][lo:hi] = v
        nbytes = sum(v.nbytes for v in obs.values() if isinstance(v, np.ndarray)) + sum(v.nbytes for v in info.values()) + done.nbytes
        if nbytes >= (8 << 20):
            # big blocks (the scans): every worker copies its own, the copies overlap.  Below that the hand-offs cost more than the
            # copies (measured: 32 768 envs over 8 handles, 4.6 MB per step: 1.24 ms serial, 1.63 ms through the workers)
            self._each(fill)
        else:
            for k in range(len(parts)):
                fill(k)
        return obs, self.timestep, done, info

    # ------------------------------------------------------------------ the env API (F110VecEnv's, over all shards)
    def reset(self, poses, env_mask=None):
        poses = np.asarray(poses, dtype=np.float64).reshape(self.num_envs, self.num_agents, 3)
        ps = self._slices(poses)
        ms = [None] * len(self.shards) if env_mask is None else self._slices(np.asarray(env_mask, dtype=bool))
        parts = self._each(lambda k, s: s.reset(ps[k], ms[k]))
        self._after_step()
        return self._assemble(parts)

    def step(self, actions):
        acts = [None] * len(self.shards) if actions is None else \
            self._slices(np.asarray(actions, dtype=np.float64).reshape(self.num_envs, self.num_agents, 2))
        if self.threaded_step:
            parts = self._each(lambda k, s: s.step(acts[k]))
        else:
            for k, s in enumerate(self.shards):      # enqueue everywhere (returns at once) ...
                s._step_device(acts[k], sync=False)
            parts = [s.step_wait() for s in self.shards]   # ... then wait everywhere
        self._after_step()
        return self._assemble(parts)

    def step_async(self, actions):
        """enqueue the step on every shard and return (gym.vector's split); step_wait() completes it"""
        acts = self._slices(np.asarray(actions, dtype=np.float64).reshape(self.num_envs, self.num_agents, 2))
        self._each(lambda k, s: s.step_async(acts[k]))

    def step_wait(self):
        parts = self._each(lambda k, s: s.step_wait())
        self._after_step()
        return self._assemble(parts)

    @property
    def action_buffers(self):
        """per shard: the page-locked [E_k][A][2] buffer its kernels read in place (fill them and call step(None))"""
        return [s.action_buffer for s in self.shards]

    def update_params(self, params, index=-1):
        self._each(lambda k, s: s.sim.update_params(params, agent_idx=index))

    def update_params_batch(self, params):
        """[E*A] dicts or [E*A][18] array, global agent order; None: back to the per-slot sets"""
        if params is None:
            self._each(lambda k, s: s.update_params_batch(None))
            return
        A = self.num_agents
        self._each(lambda k, s: s.update_params_batch(params[self.bounds[k] * A:self.bounds[k + 1] * A]))

    def update_map(self, map_path, map_ext):
        self._each(lambda k, s: s.sim.set_map(map_path, map_ext))

    def set_env_maps(self, env_map):
        ms = [None] * len(self.shards) if env_map is None else self._slices(np.asarray(env_map, dtype=np.int32))
        self._each(lambda k, s: s.set_env_maps(ms[k]))

    def device_views(self):
        """per shard: F110VecEnv.device_views() (device-resident observation of that shard's envs, on its GPU)"""
        return self._each(lambda k, s: s.device_views())

    def sync(self):
        self._each(lambda k, s: s.sim.batch.sync())

    # ------------------------------------------------------------------ optional observation gather (BASELINE configs[3])
    def _init_gather(self, f32, root):
        from .core import BatchSim
        K = len(self.shards)
        if len(set(self.shard_sizes)) != 1:
            raise ValueError("gather_obs needs equal shards (num_envs divisible by the number of devices)")
        if root is not None and not 0 <= int(root) < K:
            raise ValueError("gather_root must name a shard")
        uid = BatchSim.comm_unique_id()
        self._each(lambda k, s: s.sim.batch.comm_init(K, k, uid))   # ncclCommInitRank: all ranks at once, one thread each
        N, B = self.shard_sizes[0] * self.num_agents, self.shards[0].sim.batch.B
        self._gather = {"f32": f32, "root": None if root is None else int(root), "scans": [None] * K, "scalars": [None] * K}

        def alloc(k, s):
            recv = root is None or int(root) == k
            b = s.sim.batch
            self._gather["scans"][k] = b.device_array((K if recv else 1, N, B), np.float32 if f32 else np.float64)
            self._gather["scalars"][k] = b.device_array((K if recv else 1, 7, N))
        self._each(alloc)

    def _after_step(self):
        g = self._gather
        if g is None:
            return

        def go(k, s):
            b = s.sim.batch
            if g["f32"] or g["root"] is not None:
                b.comm_gather_obs(g["scans"][k], g["scalars"][k], f32=g["f32"], root=g["root"])
            else:
                b.comm_all_gather_obs(g["scans"][k], g["scalars"][k])
        self._each(go)

    def gathered_views(self):
        """per shard (= per device): (scans DeviceArray [shards][N_k][B], scalars DeviceArray [shards][7][N_k] in
        BatchSim.OBS_SCALARS order) — every shard's observation of the step just taken, on this shard's GPU; valid once
        the shard's stream has passed the gather (`sync()` or stream-ordered work on device_views()['stream']).
        With gather_root=k only shard k's buffers are written.  The scans are the step's; the scalar block is packed from the device
        state when the gather runs, i.e. AFTER an `auto_reset` re-seat: a finished env shows its start pose there (its `done` and the
        step's own scalars are in what step() returned)."""
        if self._gather is None:
            raise ValueError("built without gather_obs")
        return list(zip(self._gather["scans"], self._gather["scalars"]))

    # ------------------------------------------------------------------ lifetime
    def close(self):
        ws, self._workers = getattr(self, "_workers", []), []
        if ws:
            def shut(k):
                s = self.shards[k]
                if s is not None:
                    s.sim.batch.close()
            try:
                _wait([w.submit(lambda k=k: shut(k)) for k, w in enumerate(ws)])
            finally:
                for w in ws:
                    w.stop()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        import sys
        if sys.is_finalizing():     # the worker threads are daemons: at interpreter shutdown they no longer answer
            return
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
