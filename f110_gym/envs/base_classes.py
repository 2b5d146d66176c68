"""f110_gym.envs.base_classes (reference: base_classes.py:40-42, 451-630) -> f1tenth_gym_amd.sim.
RaceCar, the reference's per-agent record (:45-449), has no object here: its fields are columns of
the device-resident arrays; `Simulator.agents[i]` exposes the ones user code reads."""
from f1tenth_gym_amd.sim import Integrator, Simulator  # noqa: F401
