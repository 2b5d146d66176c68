"""f1tenth_gym_amd — MI355X-native batched F1TENTH env.step() hot path.

Python host + ctypes over a C ABI (include/f110.h) + hand-written HIP kernels for gfx950.
The names mirror the reference package f110_gym.envs so existing code ports by changing the
import:  F110Env, Simulator, Integrator, ScanSimulator2D and the free kernel functions.
"""
from .core import BatchSim, DeviceArray, DEFAULT_PARAMS  # noqa: F401
from .sim import Integrator, Simulator  # noqa: F401
from .laser import ScanSimulator2D  # noqa: F401
from .racecar import RaceCar  # noqa: F401
from .env import F110Env, F110VecEnv  # noqa: F401
from .sharded import ShardedVecEnv  # noqa: F401
from .planner import PurePursuitPlanner  # noqa: F401
from .functional import (accl_constraints, steering_constraint, vehicle_dynamics_ks, vehicle_dynamics_st, pid, func_KS, func_ST,  # noqa: F401
                         perpendicular, tripleProduct, avgPoint, indexOfFurthestPoint, support, collision, collision_multiple,
                         get_trmtx, get_vertices, get_dt, xy_2_rc, distance_transform, trace_ray, get_scan, check_ttc_jit, cross,
                         are_collinear, get_range, get_blocked_view_indices, ray_cast)

__version__ = "0.1.0"


def register_gym():
    """Register 'f110-v0' like gym/f110_gym/__init__.py:1-5 (needs gym)."""
    from gym.envs.registration import register
    register(id='f110-v0', entry_point='f1tenth_gym_amd.env:F110Env')
