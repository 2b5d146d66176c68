"""BatchSim — the thin Python owner of one libf110_hip handle (one MI355X, one HIP stream).

E independent environments x A agents are stepped in lockstep on the device.  This class only
marshals NumPy arrays to the C ABI (include/f110.h); all arithmetic happens in the HIP kernels.
The reference-compatible façades (Simulator, ScanSimulator2D, F110Env) are built on it.
"""
import ctypes as C
import os
import types
import weakref

import numpy as np

from . import _ffi
from ._ffi import check, as_f64, dptr, i32ptr

DEFAULT_PARAMS = {'mu': 1.0489, 'C_Sf': 4.718, 'C_Sr': 5.4562, 'lf': 0.15875, 'lr': 0.17145,
                  'h': 0.074, 'm': 3.74, 'I': 0.04712, 's_min': -0.4189, 's_max': 0.4189,
                  'sv_min': -3.2, 'sv_max': 3.2, 'v_switch': 7.319, 'a_max': 9.51,
                  'v_min': -5.0, 'v_max': 20.0, 'width': 0.31, 'length': 0.58}  # f110_env.py:130


def trig_tables(theta_dis):
    """laser_models.py:379-381 — NumPy-computed so the device table is bit-identical."""
    theta_arr = np.linspace(0.0, 2 * np.pi, num=theta_dis)
    return np.sin(theta_arr), np.cos(theta_arr)


def beam_tables(num_beams, fov, params):
    """RaceCar's class-level per-beam tables, base_classes.py:125-158: beam angles (built by a
    Python loop, not linspace), their cosines, and the lidar-to-body-edge distance."""
    incr = fov / (num_beams - 1)
    scan_angles = np.zeros((num_beams,))
    cosines = np.zeros((num_beams,))
    side_distances = np.zeros((num_beams,))
    dist_sides = params['width'] / 2.
    dist_fr = (params['lf'] + params['lr']) / 2.
    half_pi = np.pi / 2
    for i in range(num_beams):
        angle = -fov / 2. + i * incr
        scan_angles[i] = angle
        cosines[i] = np.cos(angle)
        with np.errstate(divide='ignore'):      # a beam at angle 0 exactly divides by -0.0: side distance -inf, as in the reference, without its warning
            to_side, to_fr = _edge_distances(angle, dist_sides, dist_fr, half_pi)
        side_distances[i] = min(to_side, to_fr)
    return scan_angles, cosines, side_distances


def _edge_distances(angle, dist_sides, dist_fr, half_pi):
    """from the lidar along a beam to the car's side and to its front / rear edge (base_classes.py:139-156)"""
    if angle > 0:
        if angle < half_pi:
            return dist_sides / np.sin(angle), dist_fr / np.cos(angle)
        return dist_sides / np.cos(angle - np.pi / 2.), dist_fr / np.sin(angle - np.pi / 2.)
    if angle > -half_pi:
        return dist_sides / np.sin(-angle), dist_fr / np.cos(-angle)
    return dist_sides / np.cos(-angle - np.pi / 2), dist_fr / np.sin(-angle - np.pi / 2)


def load_map_files(map_path, map_ext):
    """ScanSimulator2D.set_map's file handling, laser_models.py:397-416: <stem><ext> image +
    yaml with 'resolution' and 'origin'.  Returns (uint8 image top-row-first, resolution, origin)."""
    from . import mapio   # stdlib zlib + NumPy: the box needs neither PIL nor PyYAML
    map_img_path = os.path.splitext(map_path)[0] + map_ext

    def pil_image():
        from PIL import Image
        with Image.open(map_img_path) as im:
            arr = np.array(im)
        if arr.dtype == np.bool_:
            # 1-bit images: PIL hands out booleans and the reference's astype(float64) sees 0. / 1. — every cell is <= 128 and
            # becomes an obstacle (laser_models.py:399-404).  Reproduced as it is (bit parity), with a warning: the map is
            # degenerate in the reference too; convert the image to 8-bit grayscale.
            import warnings
            warnings.warn("%s is a 1-bit image: like the reference, every cell thresholds to 'occupied' (values 0 / 1 <= 128); "
                          "save the map as 8-bit grayscale" % map_img_path)
            arr = arr.astype(np.uint8)
        if arr.ndim != 2:
            raise ValueError("map image must be single-channel grayscale, got shape %s" % (arr.shape,))
        return arr
    if map_img_path.lower().endswith(".png"):
        try:
            img = mapio.read_png_gray(map_img_path)
        except ValueError as ex:   # palette / 1-2-4-bit / interlaced PNGs: what the reference hands to PIL, if PIL is here
            try:
                img = pil_image()
            except ImportError:
                raise ex
    else:   # any other image format the reference would hand to PIL: PIL it is, if installed
        img = pil_image()
    if img.dtype != np.uint8:
        # the reference thresholds the decoded values at 128 (laser_models.py:403-404)
        img = np.where(img.astype(np.float64) > 128., 255, 0).astype(np.uint8)
    try:
        meta = mapio.read_map_yaml(map_path)
    except ValueError as ex:   # block-style lists / nested keys: yaml.safe_load like the reference (:410-416), if PyYAML is here
        try:
            import yaml
        except ImportError:
            raise ex
        with open(map_path) as f:
            meta = yaml.safe_load(f)
    return np.ascontiguousarray(img), float(meta['resolution']), [float(v) for v in meta['origin']]


from . import _dlpack  # noqa: E402


class DeviceArray(object):
    """A device buffer owned by a BatchSim handle (exposes __cuda_array_interface__ so torch /
    cupy can wrap it without a copy; ROCm uses the same protocol)."""

    def __init__(self, sim, shape, dtype=np.float64, ptr=None):
        self.sim = sim
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self._owned = ptr is None
        if ptr is None:
            p = C.c_void_p()
            check(_ffi.lib().f110_device_alloc(sim._h, self.nbytes, C.byref(p)), sim._h)
            ptr = p.value
        self.ptr = ptr
        # an owned buffer is returned to the device when the array is collected, on `with` exit, or by
        # free() — whichever comes first; never after its handle is gone (the handle frees nothing of ours,
        # but f110_device_free needs it alive: BatchSim.close() runs the outstanding finalisers first)
        self._fin = None
        if self._owned:
            holder = []
            self._fin = weakref.finalize(self, DeviceArray._release, sim, ptr, holder)
            holder.append(self._fin)
            sim._device_arrays.add(self._fin)

    @staticmethod
    def _release(sim, ptr, holder=None):
        if holder:                       # the finaliser takes itself off the handle's list (no growth in long loops)
            sim._device_arrays.discard(holder[0])
        if sim._h and ptr:
            _ffi.lib().f110_device_free(sim._h, ptr)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.free()
        return False

    @property
    def __cuda_array_interface__(self):
        return {"shape": self.shape, "typestr": self.dtype.str, "data": (self.ptr, False), "version": 2}

    # ---- DLPack (the neutral zero-copy hand-off: torch.from_dlpack(arr), cupy.from_dlpack(arr), jax ... — no type of this
    # package crosses the boundary).  kDLROCM = 10; C-contiguous (strides = NULL); the capsule keeps this array alive until
    # the consumer's deleter runs.
    def __dlpack_device__(self):
        return (_dlpack.kDLROCM, int(getattr(self.sim, "device_id", 0)))

    def __dlpack__(self, stream=None, max_version=None, dl_device=None, copy=None):
        """stream = -1: no synchronisation (the caller orders its work against device_views()['stream'] itself); anything
        else: the handle's stream is drained first, so the consumer sees finished data whatever stream it uses.
        max_version / dl_device / copy: the newer protocol's keywords — a version-0 ("dltensor") capsule is what this producer
        makes whatever max_version says (consumers accept it), the data cannot leave its device and is never copied.
        Lifetime: the tensor a consumer makes of the capsule views the simulator's memory; BatchSim.close() refuses to free it
        while such a tensor is alive (drop the tensors first, or close(force=True))."""
        if self.ptr is None:
            raise ValueError("the buffer has been freed")
        if copy is True:
            raise BufferError("DeviceArray.__dlpack__ cannot copy (copy=True): it exports the simulator's own buffer")
        if dl_device is not None and tuple(int(v) for v in dl_device) != self.__dlpack_device__():
            raise BufferError("DeviceArray lives on %s and cannot be exported to dl_device=%s" % (self.__dlpack_device__(), tuple(dl_device)))
        if stream != -1 and getattr(self.sim, "_h", None):
            self.sim.sync()
        return _dlpack.make_capsule(self)

    def upload(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        if host.nbytes != self.nbytes:
            raise ValueError("size mismatch")
        check(_ffi.lib().f110_memcpy_h2d(self.sim._h, self.ptr, host.ctypes.data, self.nbytes), self.sim._h)

    def download(self):
        out = np.empty(self.shape, dtype=self.dtype)
        check(_ffi.lib().f110_memcpy_d2h(self.sim._h, out.ctypes.data, self.ptr, self.nbytes), self.sim._h)
        return out

    def download_part(self, first_row, n_rows):
        """rows [first_row, first_row + n_rows) along the leading axis (a receive buffer of many ranks' blocks is
        read back one block at a time instead of as one multi-GB host array)"""
        first_row, n_rows = int(first_row), int(n_rows)
        if first_row < 0 or n_rows < 0 or first_row + n_rows > self.shape[0]:
            raise IndexError("rows [%d, %d) are outside an array of %d rows" % (first_row, first_row + n_rows, self.shape[0]))
        row_bytes = (int(np.prod(self.shape[1:])) if len(self.shape) > 1 else 1) * self.dtype.itemsize
        out = np.empty((int(n_rows),) + self.shape[1:], dtype=self.dtype)
        check(_ffi.lib().f110_memcpy_d2h(self.sim._h, out.ctypes.data, self.ptr + int(first_row) * row_bytes, out.nbytes), self.sim._h)
        return out

    def free(self):
        if self._fin is not None:
            self.sim._device_arrays.discard(self._fin)
            self._fin()          # runs at most once
        self.ptr = None


class BatchSim(object):
    def __init__(self, params=None, num_envs=1, num_agents=2, num_beams=1080, fov=4.7, eps=0.0001,
                 theta_dis=2000, max_range=30.0, time_step=0.01, integrator=_ffi.INTEGRATOR_RK4,
                 lidar_dist=0.0, ttc_thresh=0.005, device_id=0, map_layout=_ffi.MAP_DEFAULT,
                 scan_block=0, scan_tasks_per_wave=0, step_groups=0, step_graph=0, exp=None):
        self._h = None
        self._device_arrays = set()   # finalisers of the DeviceArrays this handle owns memory for
        L = _ffi.lib()
        self.params = dict(DEFAULT_PARAMS if params is None else params)
        self.E, self.A, self.B = int(num_envs), int(num_agents), int(num_beams)
        self.N = self.E * self.A
        self.fov, self.theta_dis, self.time_step = float(fov), int(theta_dis), float(time_step)
        self.device_id = int(device_id)
        cfg = _ffi.Config()
        cfg.abi_version = _ffi.ABI_VERSION
        cfg.num_envs, cfg.num_agents, cfg.num_beams = self.E, self.A, self.B
        cfg.theta_dis, cfg.integrator, cfg.device_id = self.theta_dis, int(integrator), self.device_id
        cfg.map_layout, cfg.scan_block = int(map_layout), int(scan_block)
        cfg.scan_tasks_per_wave = int(scan_tasks_per_wave)
        cfg.step_groups = int(step_groups)
        cfg.step_graph = int(step_graph)
        cfg.fov, cfg.eps, cfg.max_range = float(fov), float(eps), float(max_range)
        cfg.time_step, cfg.lidar_dist, cfg.ttc_thresh = float(time_step), float(lidar_dist), float(ttc_thresh)
        pv = _ffi.params_vector(self.params)
        for i in range(_ffi.NPARAMS):
            cfg.params[i] = pv[i]
        h = C.c_void_p()
        check(L.f110_create(C.byref(cfg), C.byref(h)), None)
        self._h = h.value
        self._f_step_host = _ffi.lib().f110_step_host
        # NumPy-computed tables (bit-identical to the reference's)
        s, c = trig_tables(self.theta_dis)
        check(L.f110_set_trig_tables(self._h, dptr(s), dptr(c), self.theta_dis), self._h)
        self.scan_angles, self.cosines, self.side_distances = beam_tables(self.B, self.fov, self.params)
        check(L.f110_set_beam_tables(self._h, dptr(self.scan_angles), dptr(self.cosines),
                                     dptr(self.side_distances), self.B), self._h)
        self.has_map = False
        self.noise_rows = 0
        # experimental build only (libf110_hip_exp.so, F110_LIB_VARIANT=experimental): A/B switches, from the
        # `exp` argument and from F110_EXP="key=value,key=value" — read here, by the host; the library itself
        # reads no environment variable.  The product library refuses every key (ExperimentalOnly).
        switches = dict(kv.split("=", 1) for kv in os.environ.get("F110_EXP", "").split(",") if "=" in kv)
        switches.update(exp or {})
        for key, val in switches.items():
            self.exp_set(key, int(val))

    def exp_set(self, key, value):
        check(_ffi.lib().f110_exp_set(self._h, str(key).encode(), int(value)), self._h)

    # ------------------------------------------------------------------ lifetime
    def close(self, force=False):
        """frees the handle and every device buffer it owns.  Refused (BufferError) while a DLPack consumer still holds one of
        them (a torch tensor made by from_dlpack would point at freed memory): drop those tensors first, or force=True"""
        if self._h:
            if not force and _dlpack.exports_of(self):
                raise BufferError("%d DLPack export(s) of this simulator's buffers are still alive; delete the tensors made from "
                                  "them before close(), or close(force=True)" % _dlpack.exports_of(self))
            for fin in list(self._device_arrays):   # device buffers still alive: give them back first
                fin()
            self._device_arrays.clear()
            # pinned host blocks (pinned_empty) belong to the NumPy arrays that view them: each block is
            # unpinned and freed when its last view is collected, which may be after this handle is gone
            _ffi.lib().f110_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                self.close(force=True)
            except _ffi.F110LibraryError as ex:   # interpreter teardown: report, do not raise from a finaliser
                import sys
                print("BatchSim.__del__: %s" % ex, file=sys.stderr)

    def sync(self):
        check(_ffi.lib().f110_sync(self._h), self._h)

    def fence(self):
        """f110_stream_fence: call between this handle's calls and EXTERNAL work on device_views()['stream'] (torch / cupy on an
        ExternalStream around it): the handle's work in flight — both env blocks of a two-block step — is ordered in front of
        that work, and the next step runs behind it.  Never blocks the host."""
        check(_ffi.lib().f110_stream_fence(self._h), self._h)

    # ------------------------------------------------------------------ configuration
    def set_map(self, map_path, map_ext):
        img, res, origin = load_map_files(map_path, map_ext)
        self.set_map_image(img, res, origin)

    def add_map(self, map_path, map_ext):
        """register another track from its yaml + image files; returns its slot (set_env_maps)"""
        img, res, origin = load_map_files(map_path, map_ext)
        return self.add_map_image(img, res, origin)

    def set_map_image(self, img_top_first, resolution, origin):
        img = np.ascontiguousarray(img_top_first, dtype=np.uint8)
        if img.ndim != 2:
            raise ValueError("map image must be 2-D")
        check(_ffi.lib().f110_set_map_image(self._h, img.ctypes.data_as(_ffi._u8p), img.shape[0], img.shape[1],
                                            float(resolution), float(origin[0]), float(origin[1]),
                                            float(origin[2])), self._h)
        self.has_map = True
        self.map_resolution, self.map_origin = float(resolution), list(origin)

    def set_map_dt(self, dt, resolution, origin):
        dt = as_f64(dt)
        if dt.ndim != 2:
            raise ValueError("distance table must be 2-D")
        check(_ffi.lib().f110_set_map_dt(self._h, dptr(dt), dt.shape[0], dt.shape[1], float(resolution),
                                         float(origin[0]), float(origin[1]), float(np.cos(origin[2])),
                                         float(np.sin(origin[2]))), self._h)
        self.has_map = True
        self.map_resolution, self.map_origin = float(resolution), list(origin)

    # ---- a different track per env (extension; slot 0 is the map of set_map_*)
    def add_map_image(self, img_top_first, resolution, origin):
        """register another map (same pipeline as set_map_image); returns its slot"""
        img = np.ascontiguousarray(img_top_first, dtype=np.uint8)
        if img.ndim != 2:
            raise ValueError("map image must be 2-D")
        slot = C.c_int32(0)
        check(_ffi.lib().f110_add_map_image(self._h, img.ctypes.data_as(_ffi._u8p), img.shape[0], img.shape[1], float(resolution),
                                            float(origin[0]), float(origin[1]), float(origin[2]), C.byref(slot)), self._h)
        return int(slot.value)

    def add_map_dt(self, dt, resolution, origin):
        dt = as_f64(dt)
        if dt.ndim != 2:
            raise ValueError("distance table must be 2-D")
        slot = C.c_int32(0)
        check(_ffi.lib().f110_add_map_dt(self._h, dptr(dt), dt.shape[0], dt.shape[1], float(resolution), float(origin[0]),
                                         float(origin[1]), float(np.cos(origin[2])), float(np.sin(origin[2])), C.byref(slot)), self._h)
        return int(slot.value)

    def set_env_maps(self, env_map):
        """env_map [num_envs] of slots (None: every env back on slot 0)"""
        if env_map is None:
            check(_ffi.lib().f110_set_env_maps(self._h, None), self._h)
            return
        m = np.ascontiguousarray(env_map, dtype=np.int32).reshape(-1)
        if m.shape[0] != self.E:
            raise ValueError("env_map must have num_envs=%d entries (got %d)" % (self.E, m.shape[0]))
        check(_ffi.lib().f110_set_env_maps(self._h, m.ctypes.data_as(_ffi._i32p)), self._h)

    def get_map_dt(self):
        h, w = C.c_int32(), C.c_int32()
        check(_ffi.lib().f110_map_shape(self._h, C.byref(h), C.byref(w)), self._h)
        out = np.empty((h.value, w.value))
        check(_ffi.lib().f110_get_map_dt(self._h, dptr(out)), self._h)
        return out

    def set_beam_tables(self, scan_angles, cosines, side_distances):
        """replace the per-beam tables of check_ttc_jit / ray_cast (base_classes.py:125-158) — normally built from `params` at creation"""
        sa, co, sd = as_f64(scan_angles, (self.B,)), as_f64(cosines, (self.B,)), as_f64(side_distances, (self.B,))
        check(_ffi.lib().f110_set_beam_tables(self._h, dptr(sa), dptr(co), dptr(sd), self.B), self._h)
        self.scan_angles, self.cosines, self.side_distances = sa, co, sd

    def set_params(self, params, agent_idx=-1):
        pv = _ffi.params_vector(params)
        check(_ffi.lib().f110_set_params(self._h, int(agent_idx), dptr(pv)), self._h, IndexError)

    def set_params_batch(self, params):
        """one vehicle parameter set per agent: a list of N dicts or an [N][18] array in PARAM_KEYS
        order; None returns to the per-slot sets of set_params"""
        if params is None:
            check(_ffi.lib().f110_set_params_batch(self._h, None), self._h)
            return
        if isinstance(params, (list, tuple)) and len(params) and isinstance(params[0], dict):
            params = np.stack([_ffi.params_vector(p) for p in params])
        pv = as_f64(params)
        if pv.shape != (self.N, len(_ffi.PARAM_KEYS)):
            raise ValueError("per-agent parameters must be [num_envs*num_agents][%d]" % len(_ffi.PARAM_KEYS))
        check(_ffi.lib().f110_set_params_batch(self._h, dptr(pv)), self._h)

    def set_noise_table(self, noise):
        if noise is None:
            check(_ffi.lib().f110_set_noise_table(self._h, None, 0, self.B), self._h)
            self.noise_rows = 0
            return
        noise = as_f64(noise)
        if noise.ndim != 2 or noise.shape[1] != self.B:
            raise ValueError("noise table must be [rows][num_beams]")
        check(_ffi.lib().f110_set_noise_table(self._h, dptr(noise), noise.shape[0], self.B), self._h)
        self.noise_rows = noise.shape[0]

    # ---- scan noise generated on the device: np.random.default_rng(seed).normal(0, std, B) per scan
    @staticmethod
    def pcg64_state(seed):
        """(state.hi, state.lo, inc.hi, inc.lo) of np.random.PCG64(seed), computed by the library's
        restatement of SeedSequence + pcg64_set_seed (seed: int in [0, 2^64))"""
        out = (C.c_uint64 * 4)()
        check(_ffi.lib().f110_pcg64_seed(C.c_uint64(int(seed)), out), None)
        return np.array(list(out), dtype=np.uint64)

    @staticmethod
    def _state_words(seed):
        """any seed NumPy accepts (None, int of any size, sequence, SeedSequence) -> the 4 state words"""
        if isinstance(seed, (int, np.integer)) and 0 <= int(seed) < 2 ** 64:
            return BatchSim.pcg64_state(seed)
        st = np.random.PCG64(seed).state['state']
        m = (1 << 64) - 1
        return np.array([st['state'] >> 64, st['state'] & m, st['inc'] >> 64, st['inc'] & m], dtype=np.uint64)

    def set_noise_rng(self, seed, std_dev=0.01, per_agent_seeds=None, cache_rows=0, enabled=True):
        """the reference's scan noise (laser_models.py:450-452, base_classes.py:204) drawn on the
        device, bit-identical to NumPy's stream.  seed: every agent's stream (as in the reference) —
        anything np.random.default_rng accepts, None included (fresh OS entropy, exactly as
        default_rng(None) in the reference); per_agent_seeds [N]: a stream per agent instead (extension).
        enabled=False (or set_noise_off()) switches the noise off; a seed never does."""
        L = _ffi.lib()
        if not enabled:
            check(L.f110_set_noise_rng(self._h, None, 0, 0.0, 0), self._h)
        elif per_agent_seeds is not None:
            seeds = list(per_agent_seeds)
            if len(seeds) != self.N:
                raise ValueError("per_agent_seeds must have num_envs*num_agents=%d entries" % self.N)
            words = np.ascontiguousarray(np.stack([self._state_words(s) for s in seeds]), dtype=np.uint64)
            check(L.f110_set_noise_rng(self._h, words.ctypes.data_as(_ffi._u64p), 1, float(std_dev), 0), self._h)
        else:
            words = np.ascontiguousarray(self._state_words(seed), dtype=np.uint64)
            check(L.f110_set_noise_rng(self._h, words.ctypes.data_as(_ffi._u64p), 0, float(std_dev), int(cache_rows)), self._h)
        self.noise_rows = 0

    def set_noise_off(self):
        self.set_noise_rng(None, enabled=False)

    def noise_prepare(self, rows):
        check(_ffi.lib().f110_noise_prepare(self._h, int(rows)), self._h)

    def noise_rows_batch(self, seed, rows, num_beams=None, std_dev=0.01):
        """unit entry point: `rows` consecutive rng.normal(0, std_dev, num_beams) draws of
        default_rng(seed) -> (array [rows][num_beams], generator state after them as a Python int)"""
        B = self.B if num_beams is None else int(num_beams)
        words = np.ascontiguousarray(self._state_words(seed), dtype=np.uint64)
        out = np.empty((int(rows), B))
        st = (C.c_uint64 * 2)()
        check(_ffi.lib().f110_noise_rows_batch(self._h, words.ctypes.data_as(_ffi._u64p), float(std_dev), int(rows), B, dptr(out), st), self._h)
        return out, (int(st[0]) << 64) | int(st[1])

    def scan_lookup_count(self, enable=None, read=True, detail=False):
        """measurement aid: table lookups of every ray the step's scan kernels marched since the last read
        (detail=True: also how many of them the LDS window of map_layout 4 served)"""
        v = (C.c_int64 * 2)()
        check(_ffi.lib().f110_scan_lookup_count(self._h, -1 if enable is None else int(bool(enable)), v if read else None), self._h)
        return (int(v[0]), int(v[1])) if detail else int(v[0])

    # ------------------------------------------------------------------ reset / step
    def reset(self, poses, env_mask=None):
        poses = as_f64(poses)
        if poses.shape[0] != self.N:
            raise ValueError('Number of poses for reset does not match number of agents.')  # base_classes.py:625
        poses = as_f64(poses, (self.N, 3))
        mptr = None
        if env_mask is not None:
            m = np.ascontiguousarray(env_mask, dtype=np.uint8)
            if m.shape != (self.E,):
                raise ValueError("env_mask must have num_envs entries")
            mptr = m.ctypes.data_as(_ffi._u8p)
        check(_ffi.lib().f110_reset(self._h, dptr(poses), mptr), self._h)

    def step(self, actions):
        actions = as_f64(actions, (self.N, 2))
        check(_ffi.lib().f110_step(self._h, dptr(actions)), self._h)

    def step_device(self, d_actions):
        ptr = d_actions.ptr if isinstance(d_actions, DeviceArray) else int(d_actions)
        check(_ffi.lib().f110_step_device(self._h, ptr), self._h)

    def reset_device(self, d_poses, d_mask=None):
        p = d_poses.ptr if isinstance(d_poses, DeviceArray) else int(d_poses)
        m = None if d_mask is None else (d_mask.ptr if isinstance(d_mask, DeviceArray) else int(d_mask))
        check(_ffi.lib().f110_reset_device(self._h, p, m), self._h)

    def reset_collided_device(self, d_start_poses, ego_idx=0, d_count=None):
        """device-side mask reset of every env whose ego has collided (no host sync)"""
        p = d_start_poses.ptr if isinstance(d_start_poses, DeviceArray) else int(d_start_poses)
        c = None if d_count is None else (d_count.ptr if isinstance(d_count, DeviceArray) else int(d_count))
        check(_ffi.lib().f110_reset_collided_device(self._h, p, int(ego_idx), c), self._h, IndexError)

    def set_auto_reseat(self, d_start_poses, ego_idx=0, d_count=None):
        """fold reset_collided_device into the end of every following step (None disarms it)"""
        p = None if d_start_poses is None else (d_start_poses.ptr if isinstance(d_start_poses, DeviceArray) else int(d_start_poses))
        c = None if d_count is None else (d_count.ptr if isinstance(d_count, DeviceArray) else int(d_count))
        check(_ffi.lib().f110_set_auto_reseat(self._h, p, int(ego_idx), c), self._h, IndexError)
        self._reseat_refs = (d_start_poses, d_count)   # keep the device buffers alive while armed

    # ------------------------------------------------------------------ optional RCCL observation gather
    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(128)
        check(_ffi.lib().f110_comm_unique_id(buf), None)
        return buf.raw

    def comm_init(self, n_ranks, rank, unique_id):
        if len(unique_id) != 128:
            raise ValueError("unique id must be 128 bytes")
        buf = C.create_string_buffer(bytes(unique_id), 128)
        check(_ffi.lib().f110_comm_init(self._h, int(n_ranks), int(rank), buf), self._h)
        self.comm_ranks = int(n_ranks)

    def comm_set_overlap(self, enable=True):
        """double-buffer the scans so that comm_all_gather_scans overlaps the following step"""
        check(_ffi.lib().f110_comm_set_overlap(self._h, 1 if enable else 0), self._h)

    def comm_all_gather_scans(self, d_recv):
        ptr = d_recv.ptr if isinstance(d_recv, DeviceArray) else int(d_recv)
        check(_ffi.lib().f110_comm_all_gather_scans(self._h, ptr), self._h)

    OBS_SCALARS = ("poses_x", "poses_y", "poses_theta", "linear_vels_x", "linear_vels_y", "ang_vels_z", "collisions")

    def comm_all_gather_obs(self, d_recv_scans, d_recv_scalars):
        """the whole observation to every rank: scans [ranks][N][B] and the scalar block
        [ranks][7][N] in OBS_SCALARS order (one RCCL group, see f110_comm_all_gather_obs)"""
        a = d_recv_scans.ptr if isinstance(d_recv_scans, DeviceArray) else int(d_recv_scans)
        b = d_recv_scalars.ptr if isinstance(d_recv_scalars, DeviceArray) else int(d_recv_scalars)
        check(_ffi.lib().f110_comm_all_gather_obs(self._h, a, b), self._h)

    def comm_gather_obs(self, d_recv_scans, d_recv_scalars, f32=False, root=None):
        """f110_comm_gather_obs: the observation gather with float32 transport of the scans (d_recv_scans then holds
        float32 [ranks][N][B]) and / or to ONE receiving rank (root; None: every rank, the all-gather)"""
        def ptr(x):
            return None if x is None else (x.ptr if isinstance(x, DeviceArray) else int(x))
        check(_ffi.lib().f110_comm_gather_obs(self._h, ptr(d_recv_scans), ptr(d_recv_scalars), _ffi.GATHER_F32 if f32 else _ffi.GATHER_F64,
                                              -1 if root is None else int(root)), self._h)

    def step_groups(self):
        """(env blocks the handle can submit a step as, candidate streams probed at creation, blocks of the most recent
        step_device) — f110_step_groups"""
        g, p, l = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        check(_ffi.lib().f110_step_groups(self._h, C.byref(g), C.byref(p), C.byref(l)), self._h)
        return int(g.value), int(p.value), int(l.value)

    def step_launches(self):
        """1 when the most recent step ran as ONE kernel launch (k_step_tiny: at most 64 agents, one or two per env), else 0"""
        n = C.c_int32(0)
        check(_ffi.lib().f110_step_launches(self._h, C.byref(n)), self._h)
        return int(n.value)

    def comm_info(self):
        """(n_ranks, rank) as RCCL reports them"""
        n, r = C.c_int32(0), C.c_int32(-1)
        check(_ffi.lib().f110_comm_info(self._h, C.byref(n), C.byref(r)), self._h)
        return int(n.value), int(r.value)

    # ------------------------------------------------------------------ episode logic on the device
    def episode_init(self, ego_idx=0):
        check(_ffi.lib().f110_episode_init(self._h, int(ego_idx)), self._h, IndexError)
        self._ep_ego = int(ego_idx)

    def episode_reset(self, poses, env_mask=None):
        """F110Env.reset's state part (f110_env.py:319-334) for the masked envs: lap bookkeeping
        cleared, start poses + start_rot stored, simulator state reset."""
        poses = as_f64(poses, (self.N, 3))
        th = -poses.reshape(self.E, self.A, 3)[:, self._ep_ego, 2]
        rot = np.stack([np.cos(th), -np.sin(th), np.sin(th), np.cos(th)], axis=1)   # start_rot :331
        rot = np.ascontiguousarray(rot, dtype=np.float64)
        mptr = None
        if env_mask is not None:
            m = np.ascontiguousarray(env_mask, dtype=np.uint8)
            mptr = m.ctypes.data_as(_ffi._u8p)
        check(_ffi.lib().f110_episode_reset(self._h, dptr(poses), dptr(rot), mptr), self._h)

    _ep_ego = 0

    def episode_step_device(self, d_actions):
        ptr = d_actions.ptr if isinstance(d_actions, DeviceArray) else int(d_actions)
        check(_ffi.lib().f110_episode_step_device(self._h, ptr), self._h)

    def episode_reset_done_device(self, d_count=None):
        c = None if d_count is None else (d_count.ptr if isinstance(d_count, DeviceArray) else int(d_count))
        check(_ffi.lib().f110_episode_reset_done_device(self._h, c), self._h)

    def pinned_empty(self, shape, dtype=np.float64):
        """a NumPy array over page-locked host memory (full-rate DMA).  The memory lives as long as any
        array that views it (slices, .view(), reshape ...) and is returned when the last one is collected —
        closing the BatchSim first is fine."""
        shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        check(_ffi.lib().f110_host_alloc(self._h, nbytes, C.byref(p)), self._h)
        buf = (C.c_uint8 * max(nbytes, 1)).from_address(p.value)   # every NumPy view keeps `buf` alive through .base
        weakref.finalize(buf, _ffi.lib().f110_host_free, None, p.value)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    # ------------------------------------------------------------------ one call per env.step()
    HOST_FIELDS = {   # name -> (dtype, shape as a function of (N, E, B))
        "state": (np.float64, lambda N, E, B: (7, N)), "collisions": (np.float64, lambda N, E, B: (N,)),
        "collision_idx": (np.float64, lambda N, E, B: (N,)), "agent_poses": (np.float64, lambda N, E, B: (3, N)),
        "lap_times": (np.float64, lambda N, E, B: (N,)), "lap_counts": (np.float64, lambda N, E, B: (N,)),
        "toggles": (np.float64, lambda N, E, B: (N,)), "current_time": (np.float64, lambda N, E, B: (E,)),
        "in_collision": (np.int32, lambda N, E, B: (N,)), "near_starts": (np.uint8, lambda N, E, B: (N,)),
        "checkpoint_done": (np.uint8, lambda N, E, B: (N,)), "done": (np.uint8, lambda N, E, B: (E,)),
        "scans": (np.float64, lambda N, E, B: (N, B))}

    def host_block(self, fields):
        """page-locked memory for f110_step_host: returns an object with .actions ([N][2], written by the
        caller), one NumPy view per requested field (attribute .views, a dict) and the ctypes struct.
        The views are OVERWRITTEN by every step_host(): copy what you keep."""
        fields = tuple(fields)
        for f in fields:
            if f not in self.HOST_FIELDS:
                raise KeyError(f)
        N, E, B = self.N, self.E, self.B
        offs, total = {}, 16 * N     # the actions lead the block
        for f in fields:
            dt, shp = self.HOST_FIELDS[f]
            total = (total + 255) // 256 * 256
            offs[f] = total
            total += int(np.prod(shp(N, E, B))) * np.dtype(dt).itemsize
        raw = self.pinned_empty((total,), np.uint8)
        hb = types.SimpleNamespace(raw=raw, fields=fields, views={}, struct=_ffi.HostBlock())
        hb.actions = raw[:16 * N].view(np.float64).reshape(N, 2)
        hb.actions[...] = 0.0
        for f in fields:
            dt, shp = self.HOST_FIELDS[f]
            shape = shp(N, E, B)
            nb = int(np.prod(shape)) * np.dtype(dt).itemsize
            v = raw[offs[f]:offs[f] + nb].view(dt).reshape(shape)
            hb.views[f] = v
            setattr(hb.struct, f, v.ctypes.data_as(dict(_ffi.HostBlock._fields_)[f]))
        hb.struct_ref = C.byref(hb.struct)
        hb.actions_ptr = hb.actions.ctypes.data
        return hb

    def step_host_stats(self):
        """(calls, mean host us enqueuing, mean host us waiting) of step_host since the last read"""
        o = np.zeros(3)
        check(_ffi.lib().f110_step_host_stats(self._h, dptr(o)), self._h)
        n = max(o[0], 1.0)
        return int(o[0]), o[1] / n, o[2] / n

    def step_host(self, hb, actions=None, auto_reset=False, sync=True, mapped_actions=True, spin=False, fuse=True, poll=True):
        """f110_step_host: `actions` (None: hb.actions as the caller filled it in place) up, the step, the
        episode logic if episode_init was called, hb's fields down — one ABI call."""
        flags = (_ffi.STEP_AUTO_RESET if auto_reset else 0) | (0 if sync else _ffi.STEP_NO_SYNC) | (_ffi.STEP_SPIN_WAIT if spin else 0)
        if not fuse:
            flags |= _ffi.STEP_NO_FUSE
        if poll:
            flags |= _ffi.STEP_POLL
        if actions is None or actions is hb.actions:
            ptr = hb.actions_ptr
            if mapped_actions:
                flags |= _ffi.STEP_ACTIONS_MAPPED
        else:
            a = as_f64(actions, (self.N, 2))
            ptr = a.ctypes.data
        rc = _ffi.lib().f110_step_host(self._h, ptr, hb.struct_ref, flags)
        if rc:
            check(rc, self._h)

    def step_host_inplace(self, hb):
        """step_host(hb) with its defaults (the actions as the caller left them in hb.actions, read in place; wait by polling), without
        the argument handling: the single-env loop counts microseconds"""
        rc = self._f_step_host(self._h, hb.actions_ptr, hb.struct_ref, _ffi.STEP_POLL | _ffi.STEP_ACTIONS_MAPPED)
        if rc:
            check(rc, self._h)

    def episode_step_host(self, actions_pinned, packed_pinned, auto_reset=False):
        """actions up, step, _check_done, the packed scalar observation down (one copy), optional re-seat;
        returns views into packed_pinned (overwritten by the next call)"""
        check(_ffi.lib().f110_episode_step_host(self._h, actions_pinned.ctypes.data, 1 if auto_reset else 0, packed_pinned.ctypes.data), self._h)
        N, E = self.N, self.E
        cols = packed_pinned[:(9 * N + E) * 8].view(np.float64)
        fl = packed_pinned[(9 * N + E) * 8:]
        names = ("poses_x", "poses_y", "poses_theta", "linear_vels_x", "ang_vels_z", "collisions", "lap_times", "lap_counts", "toggles")
        out = {k: cols[i * N:(i + 1) * N] for i, k in enumerate(names)}
        out["current_time"] = cols[9 * N:9 * N + E]
        out["near_starts"], out["checkpoint_done"], out["done"] = fl[:N], fl[N:2 * N], fl[2 * N:2 * N + E]
        return out

    def packed_bytes(self):
        return int(_ffi.lib().f110_episode_packed_bytes(self._h))

    def episode_get(self):
        N, E = self.N, self.E
        out = {"lap_times": np.empty(N), "lap_counts": np.empty(N), "toggles": np.empty(N),
               "current_time": np.empty(E), "near_starts": np.empty(N, dtype=np.uint8),
               "done": np.empty(E, dtype=np.uint8), "checkpoint_done": np.empty(N, dtype=np.uint8)}
        o = _ffi.EpisodeHost()
        for k in ("lap_times", "lap_counts", "toggles", "current_time"):
            setattr(o, k, dptr(out[k]))
        for k in ("near_starts", "done", "checkpoint_done"):
            setattr(o, k, out[k].ctypes.data_as(_ffi._u8p))
        check(_ffi.lib().f110_episode_get(self._h, C.byref(o)), self._h)
        return out

    def episode_device_views(self):
        v = _ffi.EpisodeViews()
        check(_ffi.lib().f110_episode_device_views(self._h, C.byref(v)), self._h)
        N, E = self.N, self.E
        return {"done": DeviceArray(self, (E,), np.uint8, v.done),
                "checkpoint_done": DeviceArray(self, (N,), np.uint8, v.checkpoint_done),
                "lap_times": DeviceArray(self, (N,), np.float64, v.lap_times),
                "lap_counts": DeviceArray(self, (N,), np.float64, v.lap_counts),
                "toggles": DeviceArray(self, (N,), np.float64, v.toggles),
                "current_time": DeviceArray(self, (E,), np.float64, v.current_time)}

    def device_mem_info(self):
        """(free, total) bytes of the handle's GPU"""
        f, t = C.c_size_t(0), C.c_size_t(0)
        check(_ffi.lib().f110_device_mem_info(self._h, C.byref(f), C.byref(t)), self._h)
        return int(f.value), int(t.value)

    def device_array(self, shape, dtype=np.float64):
        return DeviceArray(self, shape, dtype)

    def device_views(self):
        v = _ffi.DeviceViews()
        check(_ffi.lib().f110_get_device_views(self._h, C.byref(v)), self._h)
        N, B = self.N, self.B
        return {"scans": DeviceArray(self, (N, B), np.float64, v.scans),
                "state": DeviceArray(self, (7, N), np.float64, v.state),
                "agent_poses": DeviceArray(self, (3, N), np.float64, v.agent_poses),
                "collisions": DeviceArray(self, (N,), np.float64, v.collisions),
                "collision_idx": DeviceArray(self, (N,), np.float64, v.collision_idx),
                "in_collision": DeviceArray(self, (N,), np.int32, v.in_collision),
                "step_count": DeviceArray(self, (N,), np.int32, v.step_count),
                "stream": v.stream}

    _F64 = ("scans", "poses_x", "poses_y", "poses_theta", "linear_vels_x", "ang_vels_z", "collisions",
            "collision_idx", "state", "agent_poses")
    _I32 = ("in_collision", "step_count")

    def get(self, *fields):
        """Read back (and synchronise).  fields from: scans poses_x poses_y poses_theta
        linear_vels_x ang_vels_z collisions collision_idx state agent_poses in_collision step_count."""
        N, B = self.N, self.B
        shapes = {"scans": (N, B), "state": (N, 7), "agent_poses": (N, 3)}
        o = _ffi.ObsHost()
        out = {}
        for f in fields:
            if f in self._F64:
                out[f] = np.empty(shapes.get(f, (N,)), dtype=np.float64)
                setattr(o, f, dptr(out[f]))
            elif f in self._I32:
                out[f] = np.empty((N,), dtype=np.int32)
                setattr(o, f, i32ptr(out[f]))
            else:
                raise KeyError(f)
        check(_ffi.lib().f110_get_obs(self._h, C.byref(o)), self._h)
        return out

    def set_state(self, state, steer_buf=None, buf_count=None):
        state = as_f64(state, (self.N, 7))
        sb = None if steer_buf is None else as_f64(steer_buf, (self.N, 2))
        bc = None if buf_count is None else np.ascontiguousarray(buf_count, dtype=np.int32)
        check(_ffi.lib().f110_set_state(self._h, dptr(state), None if sb is None else dptr(sb),
                                        None if bc is None else i32ptr(bc)), self._h)

    # ------------------------------------------------------------------ timing (bench.py)
    def timer_begin(self):
        check(_ffi.lib().f110_timer_begin(self._h), self._h)

    def timer_end_ms(self):
        ms = C.c_double()
        check(_ffi.lib().f110_timer_end_ms(self._h, C.byref(ms)), self._h)
        return ms.value

    def profile_kernels(self, enable):
        check(_ffi.lib().f110_profile_kernels(self._h, 1 if enable else 0), self._h)

    def profile_read(self):
        n = C.c_int32(); s = C.c_double(); d = C.c_double(); f = C.c_double()
        check(_ffi.lib().f110_profile_read(self._h, C.byref(n), C.byref(s), C.byref(d), C.byref(f)), self._h)
        return n.value, s.value, d.value, f.value

    # ------------------------------------------------------------------ unit entry points
    def scan_batch(self, poses, want_hits=False, want_lookups=False):
        poses = as_f64(poses)
        if poses.ndim != 2 or poses.shape[1] != 3:
            raise ValueError("poses must be [M][3]")
        m = poses.shape[0]
        ranges = np.empty((m, self.B))
        hits = np.empty((m, self.B, 2), dtype=np.int32) if want_hits else None
        lk = np.zeros((m,), dtype=np.int64) if want_lookups else None
        check(_ffi.lib().f110_scan_batch(self._h, dptr(poses), m, dptr(ranges),
                                         None if hits is None else i32ptr(hits),
                                         None if lk is None else lk.ctypes.data_as(_ffi._i64p)), self._h)
        res = [ranges]
        if want_hits:
            res.append(hits)
        if want_lookups:
            res.append(lk)
        return res[0] if len(res) == 1 else tuple(res)

    # ------------------------------------------------------------------ the reference's example policy
    def pure_pursuit_batch(self, waypoints, poses, lookahead, vgain, wheelbase, max_reacquire=20.0):
        """PurePursuitPlanner.plan (examples/waypoint_follow.py:203-217) for host poses [m][3];
        waypoints [M][3] = (x, y, speed).  Returns actions [m][2] = (steer, speed)."""
        wp = as_f64(waypoints); poses = as_f64(poses)
        if wp.ndim != 2 or wp.shape[1] != 3 or poses.ndim != 2 or poses.shape[1] != 3:
            raise ValueError("waypoints must be [M][3] = (x, y, speed) and poses [m][3]")
        out = np.empty((poses.shape[0], 2))
        check(_ffi.lib().f110_pure_pursuit_batch(self._h, dptr(wp), wp.shape[0], dptr(poses), poses.shape[0], float(lookahead),
                                                 float(vgain), float(wheelbase), float(max_reacquire), dptr(out)), self._h)
        return out

    def pure_pursuit_device(self, d_waypoints, num_waypoints, d_actions, lookahead, vgain, wheelbase, max_reacquire=20.0):
        """the same policy on the live poses of all N agents, device buffers, no host round trip"""
        w = d_waypoints.ptr if isinstance(d_waypoints, DeviceArray) else int(d_waypoints)
        a = d_actions.ptr if isinstance(d_actions, DeviceArray) else int(d_actions)
        check(_ffi.lib().f110_pure_pursuit_device(self._h, w, int(num_waypoints), float(lookahead), float(vgain), float(wheelbase),
                                                  float(max_reacquire), a), self._h)

    def scan_policy_device(self, d_actions, steer_gain=0.5, steer_max=0.4189, sector_limit=1.75, v_lo=1.0, v_hi=6.0, d_ref=6.0):
        """a reactive policy that reads the step's scans where they are (f110_scan_policy_device: steer towards the most open
        sector, speed from the clearance ahead) and writes the device action buffer — no host round trip"""
        a = d_actions.ptr if isinstance(d_actions, DeviceArray) else int(d_actions)
        check(_ffi.lib().f110_scan_policy_device(self._h, float(steer_gain), float(steer_max), float(sector_limit), float(v_lo), float(v_hi),
                                                 float(d_ref), a), self._h)

    def scan_path_stats(self, enable=None, read=True):
        """diagnostics: rays marched as dict(fast, guard, exact) since the last read; enable=True/False
        switches the counting (off by default)"""
        out = np.zeros(3, dtype=np.int64)
        check(_ffi.lib().f110_scan_path_stats(self._h, -1 if enable is None else int(bool(enable)),
                                              out.ctypes.data_as(_ffi._i64p) if read else None), self._h)
        return dict(fast=int(out[0]), guard=int(out[1]), exact=int(out[2]))

    def beam_dir_index_batch(self, thetas):
        thetas = as_f64(thetas).reshape(-1)
        idx = np.empty((thetas.shape[0], self.B), dtype=np.int32)
        check(_ffi.lib().f110_beam_dir_index_batch(self._h, dptr(thetas), thetas.shape[0], i32ptr(idx)), self._h)
        return idx

    def dynamics_batch(self, x, u, params):
        x = as_f64(x); u = as_f64(u); pv = _ffi.params_vector(params) if isinstance(params, dict) else as_f64(params, (18,))
        m = x.shape[0]
        x = as_f64(x, (m, 7)); u = as_f64(u, (m, 2))
        f_st = np.empty((m, 7)); f_ks = np.empty((m, 5))
        check(_ffi.lib().f110_dynamics_batch(self._h, dptr(x), dptr(u), dptr(pv), m, dptr(f_st), dptr(f_ks)), self._h)
        return f_st, f_ks

    def pid_batch(self, inputs, params):
        inputs = as_f64(inputs); pv = _ffi.params_vector(params) if isinstance(params, dict) else as_f64(params, (18,))
        m = inputs.shape[0]
        inputs = as_f64(inputs, (m, 4))
        out = np.empty((m, 2))
        check(_ffi.lib().f110_pid_batch(self._h, dptr(inputs), dptr(pv), m, dptr(out)), self._h)
        return out

    def update_pose_batch(self, state0, buf0, cnt0, actions, params, time_step, integrator, lidar_dist):
        state0 = as_f64(state0); m = state0.shape[0]
        state0 = as_f64(state0, (m, 7)); buf0 = as_f64(buf0, (m, 2)); actions = as_f64(actions, (m, 2))
        cnt0 = np.ascontiguousarray(cnt0, dtype=np.int32)
        pv = _ffi.params_vector(params) if isinstance(params, dict) else as_f64(params, (18,))
        s1 = np.empty((m, 7)); b1 = np.empty((m, 2)); c1 = np.empty((m,), dtype=np.int32); sp = np.empty((m, 3))
        check(_ffi.lib().f110_update_pose_batch(self._h, dptr(state0), dptr(buf0), i32ptr(cnt0), dptr(actions),
                                                dptr(pv), float(time_step), int(integrator), float(lidar_dist),
                                                m, dptr(s1), dptr(b1), i32ptr(c1), dptr(sp)), self._h, SyntaxError)
        return s1, b1, c1, sp

    def get_vertices_batch(self, poses, length, width):
        poses = as_f64(poses); m = poses.shape[0]
        poses = as_f64(poses, (m, 3))
        out = np.empty((m, 4, 2))
        check(_ffi.lib().f110_get_vertices_batch(self._h, dptr(poses), float(length), float(width), m, dptr(out)), self._h)
        return out

    def gjk_batch(self, va, vb):
        va = as_f64(va); m = va.shape[0]
        va = as_f64(va, (m, 4, 2)); vb = as_f64(vb, (m, 4, 2))
        flags = np.empty((m,), dtype=np.int32)
        check(_ffi.lib().f110_gjk_batch(self._h, dptr(va), dptr(vb), m, i32ptr(flags)), self._h)
        return flags

    def collision_multiple_batch(self, vertices):
        vertices = as_f64(vertices)
        g, n = vertices.shape[0], vertices.shape[1]
        vertices = as_f64(vertices, (g, n, 4, 2))
        col = np.empty((g, n)); idx = np.empty((g, n))
        check(_ffi.lib().f110_collision_multiple_batch(self._h, dptr(vertices), g, n, dptr(col), dptr(idx)), self._h)
        return col, idx

    def ttc_batch(self, scans, vels, ttc_thresh=0.005):
        scans = as_f64(scans); m = scans.shape[0]
        scans = as_f64(scans, (m, self.B)); vels = as_f64(vels, (m,))
        flags = np.empty((m,), dtype=np.int32)
        check(_ffi.lib().f110_ttc_batch(self._h, dptr(scans), dptr(vels), m, float(ttc_thresh), i32ptr(flags)), self._h)
        return flags

    def raycast_batch(self, ego, vertices, scans):
        ego = as_f64(ego); m = ego.shape[0]
        ego = as_f64(ego, (m, 3)); vertices = as_f64(vertices, (m, 4, 2))
        out = np.array(scans, dtype=np.float64, order='C')
        if out.shape != (m, self.B):
            raise ValueError("scans must be [M][num_beams]")
        mm = np.empty((m, 2), dtype=np.int32)
        check(_ffi.lib().f110_raycast_batch(self._h, dptr(ego), dptr(vertices), m, dptr(out), i32ptr(mm)), self._h)
        return out, mm

    def get_range_batch(self, rows):
        rows = as_f64(rows); m = rows.shape[0]
        rows = as_f64(rows, (m, 8))
        out = np.empty((m,))
        check(_ffi.lib().f110_get_range_batch(self._h, dptr(rows), m, dptr(out)), self._h)
        return out

    _HELPER_WIDTHS = {   # op -> (in width as a function of n, out width)
        _ffi.OP_ACCL_CONSTRAINTS: (lambda n: 6, 1), _ffi.OP_STEERING_CONSTRAINT: (lambda n: 6, 1), _ffi.OP_CROSS: (lambda n: 4, 1),
        _ffi.OP_ARE_COLLINEAR: (lambda n: 6, 1), _ffi.OP_PERPENDICULAR: (lambda n: 2, 2), _ffi.OP_TRIPLE_PRODUCT: (lambda n: 6, 2),
        _ffi.OP_AVG_POINT: (lambda n: 2 * n, 2), _ffi.OP_FURTHEST_POINT: (lambda n: 2 * n + 2, 1), _ffi.OP_SUPPORT: (lambda n: 4 * n + 2, 2),
        _ffi.OP_GET_TRMTX: (lambda n: 3, 16), _ffi.OP_XY_2_RC: (lambda n: 9, 2), _ffi.OP_DISTANCE_TRANSFORM: (lambda n: 2, 1),
        _ffi.OP_TRACE_RAY: (lambda n: 3, 1)}

    def helper_batch(self, op, rows, n=4):
        """f110_helper_batch: `rows` [M][in width of op] -> [M][out width] (the small star-exported functions, include/f110.h)"""
        in_w, out_w = self._HELPER_WIDTHS[op]
        rows = as_f64(rows)
        m = rows.shape[0]
        rows = as_f64(rows, (m, in_w(int(n))))
        out = np.empty((m, out_w))
        check(_ffi.lib().f110_helper_batch(self._h, int(op), dptr(rows), m, int(n), dptr(out)), self._h)
        return out

    def dt_from_bitmap(self, bitmap, resolution):
        """get_dt (laser_models.py:40-53): resolution * exact EDT of `bitmap` (nonzero = free), on the device"""
        img = np.ascontiguousarray(np.asarray(bitmap) != 0, dtype=np.uint8)
        if img.ndim != 2:
            raise ValueError("bitmap must be 2-D")
        out = np.empty(img.shape)
        check(_ffi.lib().f110_dt_from_bitmap(self._h, img.ctypes.data_as(_ffi._u8p), img.shape[0], img.shape[1], float(resolution), dptr(out)), self._h)
        return out

    def edt_sq(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        out = np.empty(img.shape, dtype=np.uint32)
        check(_ffi.lib().f110_edt_sq(self._h, img.ctypes.data_as(_ffi._u8p), img.shape[0], img.shape[1],
                                     out.ctypes.data_as(_ffi._u32p)), self._h)
        return out
