"""CPU-only tests of the Python host logic that sits above the C ABI: lap/finish bookkeeping
(f110_env.py:204-246), the shared noise table, map-file loading, integrator validation."""
import os

import numpy as np
import pytest

from _util import gold, MAPS


def test_lap_logic_matches_reference_episode():
    from f1tenth_gym_amd.env import _LapLogic
    e = gold("env_episode")
    lap = _LapLogic(1, 1, 0)
    lap.reset(e["start"].reshape(1, 1, 3))
    for k in range(len(e["x"])):
        done, tog = lap.update([e["x"][k]], [e["y"][k]], [e["col"][k]], 0.01)
        assert float(lap.toggle_list[0, 0]) == e["toggle"][k], k
        assert bool(lap.near_starts[0, 0]) == bool(e["near"][k])
        assert float(lap.lap_counts[0, 0]) == e["lap_count"][k]
        assert abs(lap.lap_times[0, 0] - e["lap_time"][k]) < 1e-12
        assert bool(done[0]) == bool(e["done"][k])
    assert bool(tog[0, 0]) and e["toggle"][-1] == 4


def test_lap_logic_single_env_scalar_form_equals_the_vectorised_one_and_the_reference():
    """_LapLogic.update_single (plain floats, what F110Env.step uses) against update() and the reference's 2-agent
    ego_idx = 1 episodes: toggles, lap counts / times, done, checkpoint flags at every step"""
    from f1tenth_gym_amd.env import _LapLogic
    g = gold("env_episode_2agents")
    for ep in range(3):
        a, b = _LapLogic(1, 2, 1), _LapLogic(1, 2, 1)
        start = g["ep%d_start" % ep].reshape(1, 2, 3)
        a.reset(start); b.reset(start)
        for k in range(len(g["ep%d_x" % ep])):
            x, y, col = g["ep%d_x" % ep][k], g["ep%d_y" % ep][k], g["ep%d_col" % ep][k]
            da, ta = a.update(x, y, col, 0.01)
            db, tb = b.update_single(list(x), list(y), col, 0.01)
            assert bool(da[0]) == db == bool(g["ep%d_done" % ep][k]), (ep, k)
            assert np.array_equal(ta[0], tb) and np.array_equal(tb, g["ep%d_ckpt" % ep][k])
            for f in ("toggle_list", "near_starts", "lap_counts", "lap_times", "current_time"):
                assert np.array_equal(getattr(a, f), getattr(b, f)), (ep, k, f)
            assert np.array_equal(b.toggle_list[0], g["ep%d_toggle" % ep][k]) and np.array_equal(b.lap_counts[0], g["ep%d_lap_count" % ep][k])
            assert np.max(np.abs(b.lap_times[0] - g["ep%d_lap_time" % ep][k])) < 1e-12


def test_lap_logic_scalar_and_vectorised_forms_agree_on_random_walks():
    """update_single (what the single-env F110Env uses) == update on random walks through and around the start zone:
    any agent count, any ego, collisions of ego and non-ego cars, start frames rotated by the ego's heading"""
    from f1tenth_gym_amd.env import _LapLogic
    rng = np.random.default_rng(5)
    for trial in range(40):
        A = int(rng.integers(1, 6)); ego = int(rng.integers(0, A))
        a, b = _LapLogic(1, A, ego), _LapLogic(1, A, ego)
        start = np.concatenate([rng.uniform(-3, 3, (1, A, 2)), rng.uniform(-3.2, 3.2, (1, A, 1))], axis=2)
        a.reset(start); b.reset(start)
        for k in range(300):
            pos = start[0, :, :2] + np.stack([1.2 * np.sin(k / 9.0 + np.arange(A)), 0.6 * np.cos(k / 13.0 + np.arange(A))], axis=1) \
                + rng.normal(0.0, 0.05, (A, 2))        # in and out of the start zone, a different phase per car
            col = (rng.random(A) < 0.01).astype(np.float64)
            da, ta = a.update(pos[:, 0], pos[:, 1], col, 0.01)
            db, tb = b.update_single(list(pos[:, 0]), list(pos[:, 1]), col, 0.01)
            assert bool(da[0]) == db and np.array_equal(ta[0], tb), (trial, k)
            for f in ("toggle_list", "near_starts", "lap_counts", "lap_times", "current_time"):
                assert np.array_equal(getattr(a, f), getattr(b, f)), (trial, k, f)
        assert a.toggle_list.max() >= 2


def test_lap_logic_vectorised_envs_are_independent():
    from f1tenth_gym_amd.env import _LapLogic
    e = gold("env_episode")
    E = 3
    lap = _LapLogic(E, 1, 0)
    starts = np.tile(e["start"].reshape(1, 1, 3), (E, 1, 1))
    starts[1, 0, :2] += 5.0
    lap.reset(starts)
    K = len(e["x"])
    for k in range(K):
        xs = np.array([e["x"][k], e["x"][k] + 5.0, e["x"][0]])
        ys = np.array([e["y"][k], e["y"][k] + 5.0, e["y"][0]])
        done, tog = lap.update(xs, ys, np.zeros(E), 0.01)
    assert list(lap.toggle_list[:, 0]) == [4, 4, 0] and list(done) == [True, True, False]
    lap.reset(starts, env_mask=[True, False, False])
    assert list(lap.toggle_list[:, 0]) == [0, 4, 0] and lap.current_time[0] == 0 and lap.current_time[1] > 0


def test_scan_noise_table_is_numpys_stream():
    """row k == the k-th successive rng.normal(0, .01, B) of a generator seeded like
    RaceCar.reset does (base_classes.py:204, laser_models.py:451); pinned by the golden row 0"""
    from f1tenth_gym_amd.sim import ScanNoise

    class FakeBatch(object):
        noise_rows = 0

        def set_noise_table(self, rows):
            self.rows = rows.copy(); self.noise_rows = rows.shape[0]
    fb = FakeBatch()
    sn = ScanNoise(12345, 1080, 0.01, chunk=4)
    sn.ensure(fb, 1)
    assert fb.noise_rows == 4
    sn.ensure(fb, 9)
    assert fb.noise_rows == 16
    rng = np.random.default_rng(seed=12345)
    for k in range(16):
        assert np.array_equal(fb.rows[k], rng.normal(0., 0.01, size=1080))
    assert np.array_equal(fb.rows[0], gold("sim_rollout")["noise_row0"])
    calls = fb.noise_rows
    sn.ensure(fb, 10)
    assert fb.noise_rows == calls


def test_map_file_loading():
    from f1tenth_gym_amd.core import load_map_files
    img, res, origin = load_map_files(os.path.join(MAPS, "example_map.yaml"), ".png")
    assert img.shape == (1600, 1600) and img.dtype == np.uint8 and res == 0.0625
    assert origin == [-78.21853769831466, -44.37590462453829, 0.0]
    with pytest.raises(FileNotFoundError):
        load_map_files(os.path.join(MAPS, "nope.yaml"), ".png")


def test_png_and_yaml_readers_need_neither_pil_nor_pyyaml(tmp_path, monkeypatch):
    """f1tenth_gym_amd.mapio (stdlib zlib + NumPy): every shipped map decodes to exactly the array PIL
    returns and the yaml fields PyYAML returns — and load_map_files works with both packages hidden"""
    import glob
    import sys
    import yaml
    from PIL import Image
    from f1tenth_gym_amd import mapio
    import f1tenth_gym_amd
    pkg_maps = os.path.join(os.path.dirname(f1tenth_gym_amd.__file__), "maps")
    pngs = sorted(glob.glob(os.path.join(MAPS, "*.png")) + glob.glob(os.path.join(pkg_maps, "*.png")))
    assert len(pngs) >= 6
    filters = set()
    for f in pngs:
        got = mapio.read_png_gray(f)
        want = np.array(Image.open(f))
        assert got.dtype == want.dtype == np.uint8 and np.array_equal(got, want), f
    for f in sorted(glob.glob(os.path.join(MAPS, "*.yaml")) + glob.glob(os.path.join(pkg_maps, "*.yaml"))):
        with open(f) as fh:
            want = yaml.safe_load(fh)
        assert mapio.read_map_yaml(f) == want, f
    # every filter type incl. Average / Paeth, 8 and 16 bit: re-encode a map with PIL's encoder variants
    src = np.array(Image.open(os.path.join(MAPS, "berlin.png")))[100:340, 150:401]
    noisy = (src.astype(np.int32) + np.random.default_rng(0).integers(-20, 20, src.shape)).clip(0, 255).astype(np.uint8)
    for k, arr in enumerate((src, noisy, noisy.astype(np.uint16) * 257)):
        out = str(tmp_path / ("t%d.png" % k))
        Image.fromarray(arr).save(out, optimize=bool(k % 2))
        assert np.array_equal(mapio.read_png_gray(out), arr)
        import struct
        import zlib
        raw = open(out, "rb").read()
        pos, idat = 8, b""
        while pos < len(raw):
            n, = struct.unpack(">I", raw[pos:pos + 4])
            if raw[pos + 4:pos + 8] == b"IDAT":
                idat += raw[pos + 8:pos + 8 + n]
            pos += 12 + n
        stride = arr.shape[1] * arr.dtype.itemsize + 1
        filters |= set(np.frombuffer(zlib.decompress(idat), np.uint8)[::stride].tolist())
    assert 4 in filters   # Paeth rows were exercised here; Average rows occur in the shipped stata_basement.png
    rgb = str(tmp_path / "rgb.png")
    Image.fromarray(np.zeros((4, 4, 3), np.uint8)).save(rgb)
    with pytest.raises(ValueError):
        mapio.read_png_gray(rgb)
    bad = tmp_path / "nested.yaml"
    bad.write_text("a:\n  b: 1\n")
    with pytest.raises(ValueError):
        mapio.read_map_yaml(str(bad))
    # what the stdlib readers refuse goes to PIL / PyYAML when they are installed (the reference's own loaders,
    # laser_models.py:397-416): a palette PNG, a 1-bit PNG, a block-style yaml list
    from f1tenth_gym_amd.core import load_map_files
    gray = np.array(Image.open(os.path.join(MAPS, "berlin.png")))[:64, :80]
    Image.fromarray(gray).convert("P").save(str(tmp_path / "pal.png"))
    Image.fromarray(gray > 128).save(str(tmp_path / "bw.png"))
    for stem in ("pal", "bw"):
        (tmp_path / (stem + ".yaml")).write_text("image: %s.png\nresolution: 0.05\norigin:\n  - -1.0\n  - 2.0\n  - 0.0\n" % stem)
        import warnings
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            img, res, origin = load_map_files(str(tmp_path / (stem + ".yaml")), ".png")
        assert img.dtype == np.uint8 and img.shape == (64, 80) and res == 0.05 and origin == [-1.0, 2.0, 0.0]
        if stem == "bw":
            # a 1-bit image is what it is in the reference: np.array(img).astype(float64) holds 0. / 1., all <= 128 -> every cell
            # an obstacle (laser_models.py:399-404) — reproduced, with a warning
            ref_like = np.array(Image.open(str(tmp_path / "bw.png"))).astype(np.float64)
            assert np.array_equal(img > 128, ref_like > 128.) and not (img > 128).any() and any("1-bit" in str(w.message) for w in caught)
    # with PIL and yaml unimportable, set_map's file handling still works
    from f1tenth_gym_amd.core import load_map_files
    monkeypatch.setitem(sys.modules, "PIL", None)
    monkeypatch.setitem(sys.modules, "yaml", None)
    img, res, origin = load_map_files(os.path.join(pkg_maps, "vegas.yaml"), ".png")
    assert img.shape == (2248, 3000) and res == 0.05 and origin == [-11.60654, -27.320793, 0.0]


def test_integrator_enum_and_validation():
    from f1tenth_gym_amd.sim import Integrator, _integrator_code
    assert Integrator.RK4.value == 1 and Integrator.Euler.value == 2
    assert _integrator_code(Integrator.Euler) == 2 and _integrator_code(1) == 1
    with pytest.raises(SyntaxError):
        _integrator_code("Heun")


def test_named_maps_ship_with_the_package():
    from f1tenth_gym_amd.env import _resolve_map_path
    for name in ("berlin", "skirk", "vegas"):
        n, path = _resolve_map_path({'map': name})
        assert os.path.isfile(path) and os.path.isfile(path[:-5] + ".png")
    n, path = _resolve_map_path({})
    assert n == 'vegas' and os.path.isfile(path)
    assert _resolve_map_path({'map': '/x/custom'})[1] == '/x/custom.yaml'


def test_f110_gym_alias_package_registers_and_exports(monkeypatch):
    """import-level drop-in: `import f110_gym` registers 'f110-v0' with gym (a stub stands in for gym,
    absent from the image), and the reference's import paths resolve to the MI355X classes"""
    import sys
    import types
    calls = []
    gym = types.ModuleType("gym")
    gym.Env = object
    gym.envs = types.ModuleType("gym.envs")
    gym.envs.registration = types.ModuleType("gym.envs.registration")
    gym.envs.registration.register = lambda **kw: calls.append(kw)
    for name, mod in (("gym", gym), ("gym.envs", gym.envs), ("gym.envs.registration", gym.envs.registration)):
        monkeypatch.setitem(sys.modules, name, mod)
    for name in [m for m in sys.modules if m == "f110_gym" or m.startswith("f110_gym.")]:
        monkeypatch.delitem(sys.modules, name)
    import f110_gym
    assert calls == [{"id": "f110-v0", "entry_point": "f110_gym.envs:F110Env"}]     # gym/f110_gym/__init__.py:1-5
    from f110_gym.envs import F110Env, Simulator, ScanSimulator2D, Integrator, pid, collision_multiple, ray_cast  # noqa: F401
    from f110_gym.envs.base_classes import RaceCar   # base_classes.py:45: importable, subclassable (VERDICT r4 missing 5)
    assert isinstance(RaceCar, type) and RaceCar.scan_simulator is None and {'update_pose', 'check_ttc', 'ray_cast_agents', 'update_scan', 'update_opp_poses', 'reset', 'set_map', 'update_params'} <= set(dir(RaceCar))
    from f110_gym.envs.base_classes import Integrator as I2
    from f110_gym.envs.f110_env import F110Env as E2
    import f1tenth_gym_amd
    assert E2 is f1tenth_gym_amd.F110Env and I2 is f1tenth_gym_amd.Integrator and I2.RK4.value == 1
    import importlib
    mod, cls = f110_gym.ENTRY_POINT.split(":")
    assert getattr(importlib.import_module(mod), cls) is f1tenth_gym_amd.F110Env      # what gym.make resolves
    import pytest
    with pytest.raises(ValueError):
        f110_gym.make("other-v0")


def test_dlpack_capsule_fields():
    """DeviceArray.__dlpack__ (f1tenth_gym_amd/_dlpack.py): a "dltensor" capsule around a DLManagedTensor with the HIP pointer,
    device (kDLROCM = 10, device id), dtype, shape, NULL strides (C-contiguous); the array stays alive until the deleter runs"""
    import ctypes as C
    import gc
    from f1tenth_gym_amd import _dlpack
    from f1tenth_gym_amd.core import DeviceArray

    class FakeSim(object):
        _h = None
        device_id = 3
        _device_arrays = set()
    for shape, dtype, code in (((5, 1080), np.float64, (2, 64, 1)), ((7,), np.int32, (0, 32, 1)), ((2, 3, 4), np.uint8, (1, 8, 1)), ((9, 2), np.float32, (2, 32, 1))):
        arr = DeviceArray(FakeSim(), shape, dtype, ptr=0x7f0000001000)
        assert arr.__dlpack_device__() == (10, 3)
        n0 = len(_dlpack._live)
        cap = arr.__dlpack__()
        f = _dlpack.read_capsule(cap)
        assert f["data"] == 0x7f0000001000 and f["device"] == (10, 3) and f["ndim"] == len(shape) and f["shape"] == shape
        assert f["dtype"] == code and f["strides"] is None and f["byte_offset"] == 0 and len(_dlpack._live) == n0 + 1
        # a consumer renames the capsule and calls the deleter when it is done with the memory
        C.pythonapi.PyCapsule_SetName.argtypes = [C.py_object, C.c_char_p]
        C.pythonapi.PyCapsule_SetName(cap, b"used_dltensor")
        mt = _dlpack.DLManagedTensor.from_address(f["address"])
        mt.deleter(C.pointer(mt))
        assert len(_dlpack._live) == n0
        del cap
        # ... and a capsule nobody consumed releases its tensor when it is collected
        cap = arr.__dlpack__(stream=-1)
        assert len(_dlpack._live) == n0 + 1
        del cap
        gc.collect()
        assert len(_dlpack._live) == n0
    # the newer protocol keywords (numpy >= 2.1, array-api callers): accepted; a copy or a foreign device is refused, not a TypeError
    sim = FakeSim()
    arr = DeviceArray(sim, (4,), np.float64, ptr=0x7f0000002000)
    cap = arr.__dlpack__(stream=-1, max_version=(1, 0), dl_device=(10, 3), copy=False)
    assert _dlpack.read_capsule(cap)["data"] == 0x7f0000002000 and _dlpack.exports_of(sim) == 1
    for kw in ({"copy": True}, {"dl_device": (1, 0)}, {"dl_device": (10, 0)}):
        with pytest.raises(BufferError):
            arr.__dlpack__(stream=-1, **kw)
    del cap
    gc.collect()
    assert _dlpack.exports_of(sim) == 0
    arr.ptr = None
    with pytest.raises(ValueError):
        arr.__dlpack__()


def test_workload_module_equals_the_test_helpers():
    """f1tenth_gym_amd/workload.py (what bench.py runs on: shipped example track, raceline, start poses, action sets) is the same data
    the tests' own helpers read from tests/golden/maps (PIL / PyYAML there, the package's own PNG / yaml readers here)"""
    from f1tenth_gym_amd import workload
    from _util import bench_start_poses, load_map_image, raceline
    img, res, origin = load_map_image("example_map")
    img2, res2, origin2 = workload.load_map_image("example_map")
    assert np.array_equal(img, img2) and res == res2 and origin == origin2
    assert np.array_equal(raceline(), workload.raceline())
    assert np.array_equal(bench_start_poses(1000, 2), workload.bench_start_poses(1000, 2))
    assert np.array_equal(bench_start_poses(50, 3, gap_wp=6), workload.start_poses(np.arange(50), 3, 6))
    assert np.array_equal(workload.start_poses(workload.shard_envs(16, 3), 2), workload.bench_start_poses(64, 2)[96:128])


def test_sharded_vec_env_rejects_bad_shardings():
    """ShardedVecEnv validates the sharding before it touches a device"""
    from f1tenth_gym_amd import ShardedVecEnv
    for kw in (dict(devices=[]), dict(devices=[0, 0], shard_sizes=[3, 3]), dict(devices=[0, 0], shard_sizes=[8, 0]), dict(devices=[0], shard_sizes=[4, 4])):
        with pytest.raises(ValueError):
            ShardedVecEnv(8, **kw)


def test_every_script_of_the_repo_compiles():
    """tools/, examples/, oracle/refshim/, bench.py, __graft_entry__.py: syntax only (the GPU-box session scripts and the fuzzers
    run where no test here can run them; a typo in one would cost a gpurun call)"""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")]
    for sub in ("tools", os.path.join("tools", "debug"), "examples", os.path.join("oracle", "refshim")):
        files += sorted(glob.glob(os.path.join(root, sub, "*.py")))
    assert len(files) > 20
    for f in files:
        with open(f) as fh:
            compile(fh.read(), f, "exec")


def test_alias_package_exports_every_public_name_of_the_reference():
    """`from f110_gym.envs import *` in the reference (envs/__init__.py:1-5) star-imports dynamic_models, laser_models, base_classes and
    collision_models: every public function / class those modules define (their unittest classes and script mains aside) must come out of
    the alias package too.  Build container only (reads the reference's module sources for their NAMES)."""
    import ast
    ref = "/root/reference/gym/f110_gym/envs"
    if not os.path.isdir(ref):
        pytest.skip("the reference tree is not here")
    want = {"F110Env"}
    for mod in ("dynamic_models", "laser_models", "collision_models", "base_classes"):
        with open(os.path.join(ref, mod + ".py")) as f:
            tree = ast.parse(f.read())
        for node in tree.body:
            if isinstance(node, ast.FunctionDef) and not node.name.startswith("_") and node.name != "main":
                want.add(node.name)
            elif isinstance(node, ast.ClassDef) and not any(getattr(b, "attr", getattr(b, "id", "")) == "TestCase" for b in node.bases):
                want.add(node.name)
    ns = {}
    exec("from f110_gym.envs import *", ns)
    assert not (want - set(ns)), sorted(want - set(ns))
    assert len(want) >= 32
