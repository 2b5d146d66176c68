"""Shared helpers for the test-suite (fixtures, map loading through the ORACLE)."""
import functools
import os

import numpy as np
import yaml
from PIL import Image

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MAPS = os.path.join(GOLD, "maps")


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


@functools.lru_cache(maxsize=None)
def load_map_image(name):
    """returns (img uint8 [H,W] top-row-first as PIL decodes it, resolution, origin[3])"""
    with open(os.path.join(MAPS, name + ".yaml")) as f:
        meta = yaml.safe_load(f)
    img = np.array(Image.open(os.path.join(MAPS, name + ".png")))
    assert img.ndim == 2 and img.dtype == np.uint8
    return img, float(meta['resolution']), [float(v) for v in meta['origin']]


@functools.lru_cache(maxsize=None)
def oracle_map_dt(name):
    """distance table of a fixture map computed by the ORACLE's EDT (pinned against scipy in
    test_oracle_golden.py)."""
    from oracle import orc
    img, res, origin = load_map_image(name)
    return orc.map_dt_from_image(img, res), res, origin


def raceline():
    return np.loadtxt(os.path.join(MAPS, "example_waypoints.csv"), delimiter=';', skiprows=3)


def bench_start_poses(num_envs, num_agents=2, gap_wp=10):
    """SURVEY §8d start poses: env e -> waypoint (e*7919) mod 783, heading psi+pi/2;
    opponent(s) gap_wp waypoints behind along the raceline."""
    w = raceline()
    n = w.shape[0]
    poses = np.empty((num_envs, num_agents, 3))
    for a in range(num_agents):
        k = ((np.arange(num_envs) * 7919) % n - a * gap_wp) % n
        poses[:, a, 0] = w[k, 1]
        poses[:, a, 1] = w[k, 2]
        poses[:, a, 2] = w[k, 3] + np.pi / 2
    return poses.reshape(num_envs * num_agents, 3)


def rel_err(a, b, atol=1e-12):
    """north_star's tolerance as a number: the smallest tol with |a - b| <= tol*|b| + atol everywhere, so
    `rel_err(a, b) < 1e-5` IS "within 1e-5 relative" — relative to the reference value itself, also below 1
    (a 0.1 m range off by 1e-6 m fails a 1e-5 gate).  The absolute floor only forgives differences of the
    size of float64 rounding around zero (atol = 1e-12; a reference of exactly 0 allows nothing more)."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    if not a.size:
        return 0.0
    a, b = np.broadcast_arrays(a, b)
    excess = np.maximum(np.abs(a - b) - atol, 0.0)
    mag = np.abs(b)
    with np.errstate(divide="ignore", invalid="ignore"):
        e = np.where(excess > 0.0, excess / mag, 0.0)   # excess > 0 over |b| == 0 -> inf
    return float(np.max(e))


PKG_MAPS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "f1tenth_gym_amd", "maps")


def map_stem(name):
    """path without extension of a track's yaml / png pair: tests/golden/maps, else the package's own maps (vegas ...)"""
    for d in (MAPS, PKG_MAPS):
        if os.path.isfile(os.path.join(d, name + ".yaml")) and os.path.isfile(os.path.join(d, name + ".png")):
            return os.path.join(d, name)
    raise FileNotFoundError(name)


@functools.lru_cache(maxsize=None)
def load_any_map_image(name):
    """like load_map_image, for any track map_stem finds"""
    stem = map_stem(name)
    with open(stem + ".yaml") as f:
        meta = yaml.safe_load(f)
    img = np.array(Image.open(stem + ".png"))
    assert img.ndim == 2 and img.dtype == np.uint8
    return img, float(meta['resolution']), [float(v) for v in meta['origin']]


def write_variant_yaml(tmp_dir, name, resolution, origin):
    """yaml + png pair under tmp_dir: the track's image under ANOTHER resolution / origin (yaw included), as
    oracle/refshim/gen_golden.py wrote it for the reference (scan_rotated_origin.npz stores the numbers); -> yaml path"""
    import shutil
    stem = os.path.join(str(tmp_dir), "%s_variant" % name)
    shutil.copyfile(map_stem(name) + ".png", stem + ".png")
    with open(stem + ".yaml", "w") as f:
        f.write("image: %s.png\nresolution: %r\norigin: [%r, %r, %r]\nnegate: 0\noccupied_thresh: 0.65\nfree_thresh: 0.196\n"
                % (os.path.basename(stem), float(resolution), float(origin[0]), float(origin[1]), float(origin[2])))
    return stem + ".yaml"
