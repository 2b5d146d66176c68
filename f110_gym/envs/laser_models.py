"""f110_gym.envs.laser_models (reference: laser_models.py:40-457) -> f1tenth_gym_amd"""
from f1tenth_gym_amd.laser import ScanSimulator2D  # noqa: F401
from f1tenth_gym_amd.functional import (get_dt, xy_2_rc, distance_transform, trace_ray, get_scan, check_ttc_jit, cross,  # noqa: F401
                                        are_collinear, get_range, get_blocked_view_indices, ray_cast)
