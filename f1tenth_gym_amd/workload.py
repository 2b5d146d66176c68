"""The benchmark workload of SURVEY §8d as product code: the example track (the reference's
examples/example_map.{png,yaml} + example_waypoints.csv, shipped under f1tenth_gym_amd/maps), the
raceline start poses and the pre-drawn action sets that bench.py, examples/ and the tests share.
Nothing here touches the oracle or the test tree: bench.py runs from a checkout without tests/."""
import functools
import os

import numpy as np

from . import mapio

PKG_MAPS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "maps")


def map_stem(name="example_map"):
    """path without extension of a shipped track's yaml / png pair"""
    stem = os.path.join(PKG_MAPS, name)
    if not (os.path.isfile(stem + ".yaml") and os.path.isfile(stem + ".png")):
        raise FileNotFoundError("no shipped track %r under %s" % (name, PKG_MAPS))
    return stem


@functools.lru_cache(maxsize=None)
def load_map_image(name="example_map"):
    """(uint8 image [H][W] top row first as PIL decodes it, resolution, origin[3]) of a shipped track —
    what ScanSimulator2D.set_map reads (laser_models.py:397-416), without PIL / PyYAML"""
    stem = map_stem(name)
    img = mapio.read_png_gray(stem + ".png")
    meta = mapio.read_map_yaml(stem + ".yaml")
    return np.ascontiguousarray(img), float(meta['resolution']), [float(v) for v in meta['origin']]


@functools.lru_cache(maxsize=None)
def raceline():
    """examples/example_waypoints.csv: columns s_m; x_m; y_m; psi_rad; kappa_radpm; vx_mps; ax_mps2
    (config_example_map.yaml:16-22); 783 points 0.2 m apart, a closed loop"""
    w = np.loadtxt(os.path.join(PKG_MAPS, "example_waypoints.csv"), delimiter=';', skiprows=3)
    w.setflags(write=False)
    return w


def shard_envs(envs_per_rank, rank):
    """global env ids owned by `rank`: contiguous blocks"""
    return np.arange(int(envs_per_rank), dtype=np.int64) + int(rank) * int(envs_per_rank)


def start_poses(env_ids, num_agents=2, gap_wp=10, order=""):
    """env e starts its ego on raceline waypoint (e * 7919) mod 783 with heading psi + pi/2 (the csv
    measures psi from +y), agent a `a * gap_wp` waypoints (0.2 m each) behind along the raceline.
    -> [len(env_ids) * num_agents][3].  order="sorted" lays the envs out along the track (an experiment)."""
    env_ids = np.asarray(env_ids, dtype=np.int64)
    w = raceline()
    n = w.shape[0]
    poses = np.empty((len(env_ids), num_agents, 3))
    for a in range(num_agents):
        k = ((env_ids * 7919) % n - a * gap_wp) % n
        if order == "sorted":
            k = (np.sort((env_ids * 7919) % n) - a * gap_wp) % n
        poses[:, a, 0] = w[k, 1]
        poses[:, a, 1] = w[k, 2]
        poses[:, a, 2] = w[k, 3] + np.pi / 2
    return poses.reshape(len(env_ids) * num_agents, 3)


def bench_start_poses(num_envs, num_agents=2, gap_wp=10):
    """start_poses of envs 0 .. num_envs-1"""
    return start_poses(np.arange(int(num_envs)), num_agents, gap_wp)


def action_sets(n_sets, n_agents, seed):
    """SURVEY §8d actions: steer ~ U(-0.2, 0.2), speed ~ U(2, 6), one set per 20 steps"""
    rng = np.random.default_rng(seed)
    return [np.stack([rng.uniform(-0.2, 0.2, n_agents), rng.uniform(2.0, 6.0, n_agents)], axis=1) for _ in range(n_sets)]
