"""Load the reference's four hot-path modules (and optionally F110Env) from /root/reference.

TEST INFRASTRUCTURE — runs only in the build container (the reference tree does not exist on
the GPU box).  It is used to (i) pin oracle/f110_oracle.c against the real reference and
(ii) generate the golden vectors under tests/golden/ (see gen_golden.py).

Recipe: a no-op `numba` shim package, empty `f110_gym` / `f110_gym.envs`
package objects in sys.modules, then importlib-load dynamic_models.py, laser_models.py,
collision_models.py, base_classes.py by path.  For F110Env, 10-line stubs of gym / pyglet.
"""
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("F110_REFERENCE_ROOT", "/root/reference")
_ENV_DIR = os.path.join(REF_ROOT, "gym", "f110_gym", "envs")
_loaded = {}


def numba_in_use():
    """what `numba` the loaded reference modules were decorated with: "real numba x.y" or "no-op shim ..." """
    return _loaded.get("numba", "not loaded yet")


def reference_available():
    return os.path.isfile(os.path.join(_ENV_DIR, "laser_models.py"))


def _load(name, filename):
    full = "f110_gym.envs." + name
    spec = importlib.util.spec_from_file_location(full, os.path.join(_ENV_DIR, filename))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[full] = mod
    spec.loader.exec_module(mod)
    return mod


def _install_gym_pyglet_stubs():
    if "gym" not in sys.modules:
        gym = types.ModuleType("gym")

        class Env(object):
            pass

        gym.Env = Env
        gym.error = types.ModuleType("gym.error")
        gym.spaces = types.ModuleType("gym.spaces")
        gym.utils = types.ModuleType("gym.utils")
        gym.utils.seeding = types.ModuleType("gym.utils.seeding")
        for sub in ("error", "spaces", "utils"):
            sys.modules["gym." + sub] = getattr(gym, sub)
        sys.modules["gym.utils.seeding"] = gym.utils.seeding
        sys.modules["gym"] = gym
    if "pyglet" not in sys.modules:
        pyglet = types.ModuleType("pyglet")
        pyglet.options = {}
        pyglet.gl = types.ModuleType("pyglet.gl")
        sys.modules["pyglet"] = pyglet
        sys.modules["pyglet.gl"] = pyglet.gl


def load_reference(with_env=False):
    """Returns a namespace with .dynamic_models .laser_models .collision_models .base_classes
    (and .f110_env when with_env=True)."""
    if not reference_available():
        raise RuntimeError("reference tree not found at %s" % REF_ROOT)
    if "core" not in _loaded:
        shim_dir = os.path.dirname(os.path.abspath(__file__))
        if "numba" not in sys.modules and os.environ.get("F110_REAL_NUMBA"):
            # gen_golden.py --check on a machine that HAS numba: the reference's @njit functions run compiled, as its users run them
            hidden = [p for p in sys.path if os.path.abspath(p or ".") == shim_dir]   # (the shim's own directory may be on the path)
            for p in hidden:
                sys.path.remove(p)
            try:
                import numba  # noqa: F401
                _loaded["numba"] = "real numba %s" % getattr(numba, "__version__", "?")
            except Exception as ex:  # noqa: BLE001 - absent, or broken against this NumPy
                sys.modules.pop("numba", None)
                _loaded["numba"] = "no-op shim (real numba is not importable here: %s: %s)" % (type(ex).__name__, str(ex)[:80])
            finally:
                sys.path[:0] = hidden
        if "numba" not in sys.modules:
            sys.path.insert(0, shim_dir)
            import numba  # noqa: F401  (the shim)
            sys.path.remove(shim_dir)
            _loaded.setdefault("numba", "no-op shim")
        pkg = types.ModuleType("f110_gym")
        pkg.__path__ = []
        envs = types.ModuleType("f110_gym.envs")
        envs.__path__ = []
        # the repo ships an `f110_gym` alias package of its own (the drop-in): the reference's
        # modules must not be mixed with it, so the names are (re)bound to empty packages here
        # ... for the duration of the load only: afterwards `import f110_gym` resolves to whatever it did before
        # (the drop-in alias, if it was imported, else a fresh import), and the reference's modules stay
        # reachable through the returned namespace alone
        saved = {k: sys.modules.get(k) for k in list(sys.modules) if k == "f110_gym" or k.startswith("f110_gym.")}
        for k in saved:
            del sys.modules[k]
        sys.modules["f110_gym"] = pkg
        sys.modules["f110_gym.envs"] = envs
        try:
            ns = types.SimpleNamespace()
            ns.dynamic_models = _load("dynamic_models", "dynamic_models.py")
            ns.laser_models = _load("laser_models", "laser_models.py")
            ns.collision_models = _load("collision_models", "collision_models.py")
            ns.base_classes = _load("base_classes", "base_classes.py")
            _loaded["core"] = ns
            _loaded["ref_modules"] = {k: v for k, v in sys.modules.items() if k == "f110_gym" or k.startswith("f110_gym.")}
        finally:
            for k in [k for k in sys.modules if k == "f110_gym" or k.startswith("f110_gym.")]:
                del sys.modules[k]
            sys.modules.update({k: v for k, v in saved.items() if v is not None})
    ns = _loaded["core"]
    if with_env and not hasattr(ns, "f110_env"):
        # f110_env.py imports its siblings by name: put the reference's modules back for the load
        saved = {k: sys.modules.get(k) for k in list(sys.modules) if k == "f110_gym" or k.startswith("f110_gym.")}
        for k in saved:
            del sys.modules[k]
        sys.modules.update(_loaded["ref_modules"])
        try:
            _install_gym_pyglet_stubs()
            ns.f110_env = _load("f110_env", "f110_env.py")
            _loaded["ref_modules"]["f110_gym.envs.f110_env"] = ns.f110_env
        finally:
            for k in [k for k in sys.modules if k == "f110_gym" or k.startswith("f110_gym.")]:
                del sys.modules[k]
            sys.modules.update({k: v for k, v in saved.items() if v is not None})
            if saved.get("f110_gym.envs.base_classes") is ns.base_classes and hasattr(ns, "f110_env"):
                sys.modules["f110_gym.envs.f110_env"] = ns.f110_env   # called inside a reference_modules() block
    return ns


class reference_modules(object):
    """`with reference_modules():` — inside the block `import f110_gym...` resolves to the REFERENCE's
    modules (a generator script that loads the reference's examples needs that); on exit the names are
    bound again to what they were before (the repo's drop-in alias package, or nothing)."""

    def __enter__(self):
        load_reference()
        self.saved = {k: sys.modules.get(k) for k in list(sys.modules) if k == "f110_gym" or k.startswith("f110_gym.")}
        for k in self.saved:
            del sys.modules[k]
        sys.modules.update(_loaded["ref_modules"])
        return _loaded["core"]

    def __exit__(self, *exc):
        for k in [k for k in sys.modules if k == "f110_gym" or k.startswith("f110_gym.")]:
            if k not in _loaded["ref_modules"]:
                _loaded["ref_modules"][k] = sys.modules[k]   # e.g. f110_env, loaded inside the block
            del sys.modules[k]
        sys.modules.update({k: v for k, v in self.saved.items() if v is not None})
        return False


def fresh_racecar_class(ns):
    """The reference keeps the scan simulator and TTC tables as RaceCar class attributes
    (base_classes.py:64-67); clear them so a new Simulator rebuilds them."""
    rc = ns.base_classes.RaceCar
    rc.scan_simulator = None
    rc.cosines = None
    rc.scan_angles = None
    rc.side_distances = None
