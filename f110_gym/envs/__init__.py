"""The names `f110_gym.envs` exports in the reference (its envs/__init__.py:1-5 star-imports the four kernel modules next
to F110Env), bound to the MI355X implementations — listed one by one here, so that `from f110_gym.envs import *` yields
exactly the reference's public names (plus F110VecEnv / ShardedVecEnv, which have no reference counterpart)."""
from f110_gym.envs.f110_env import F110Env, F110VecEnv, ShardedVecEnv  # noqa: F401
from f110_gym.envs.base_classes import Integrator, RaceCar, Simulator  # noqa: F401
from f110_gym.envs.dynamic_models import (accl_constraints, steering_constraint, vehicle_dynamics_ks, vehicle_dynamics_st, pid,  # noqa: F401
                                          func_KS, func_ST)
from f110_gym.envs.laser_models import (ScanSimulator2D, get_dt, xy_2_rc, distance_transform, trace_ray, get_scan, check_ttc_jit,  # noqa: F401
                                        cross, are_collinear, get_range, get_blocked_view_indices, ray_cast)
from f110_gym.envs.collision_models import (perpendicular, tripleProduct, avgPoint, indexOfFurthestPoint, support, collision,  # noqa: F401
                                            collision_multiple, get_trmtx, get_vertices)
