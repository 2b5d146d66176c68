"""f110_gym.envs.dynamic_models (reference: dynamic_models.py:29-229) -> f1tenth_gym_amd.functional"""
from f1tenth_gym_amd.functional import (accl_constraints, steering_constraint, vehicle_dynamics_ks, vehicle_dynamics_st, pid,  # noqa: F401
                                        func_KS, func_ST)
