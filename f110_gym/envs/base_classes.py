"""f110_gym.envs.base_classes (reference: base_classes.py:40-42, 45-449, 451-630) -> f1tenth_gym_amd.sim / .racecar.
`RaceCar` is the reference's per-vehicle class with its constructor, attributes and methods (unit entry points, one launch per
call); inside `Simulator` the per-agent records are columns of device-resident arrays and `Simulator.agents[i]` hands out views."""
from f1tenth_gym_amd.sim import Integrator, Simulator  # noqa: F401
from f1tenth_gym_amd.racecar import RaceCar  # noqa: F401
