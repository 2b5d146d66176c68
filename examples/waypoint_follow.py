"""BASELINE configs[0] on the MI355X path: one car, example_map, the pure-pursuit example policy.

The counterpart of the reference's examples/waypoint_follow.py with the imports switched:
`F110Env` and `PurePursuitPlanner` come from f1tenth_gym_amd (both run on the GPU), rendering is
left out.  Runs until the env reports done (two laps) and prints the lap time.

    python examples/waypoint_follow.py
"""
import os
import sys
import time
from argparse import Namespace

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from f1tenth_gym_amd import F110Env, Integrator, PurePursuitPlanner  # noqa: E402


def load_conf(path=os.path.join(HERE, "run_example_map.yaml")):
    with open(path) as f:
        conf = Namespace(**yaml.safe_load(f))
    for key in ("map_path", "wpt_path"):     # paths in the config are relative to the config file
        setattr(conf, key, os.path.normpath(os.path.join(os.path.dirname(path), getattr(conf, key))))
    return conf


def run(conf, lookahead=0.82461887897713965, vgain=1.375, max_steps=20000):
    planner = PurePursuitPlanner(conf, 0.17145 + 0.15875)
    env = F110Env(map=conf.map_path, map_ext=conf.map_ext, num_agents=1, timestep=0.01, integrator=Integrator.RK4)
    obs, step_reward, done, info = env.reset(np.array([[conf.sx, conf.sy, conf.stheta]]))
    laptime, steps = 0.0, 0
    while not done and steps < max_steps:
        speed, steer = planner.plan(obs['poses_x'][0], obs['poses_y'][0], obs['poses_theta'][0], lookahead, vgain)
        obs, step_reward, done, info = env.step(np.array([[steer, speed]]))
        laptime += step_reward
        steps += 1
    planner.close()
    return dict(steps=steps, laptime=laptime, lap_count=float(obs['lap_counts'][0]), lap_time=float(obs['lap_times'][0]),
                collided=bool(obs['collisions'][0]), done=bool(done))


if __name__ == '__main__':
    t0 = time.time()
    res = run(load_conf())
    print('Sim elapsed time:', res['laptime'], 'Real elapsed time:', time.time() - t0, res)
