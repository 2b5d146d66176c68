"""f110_gym.envs.dynamic_models (reference: dynamic_models.py:90-221) -> f1tenth_gym_amd.functional"""
from f1tenth_gym_amd.functional import vehicle_dynamics_st, vehicle_dynamics_ks, pid  # noqa: F401
