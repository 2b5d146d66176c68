"""f110_gym.envs.laser_models (reference: laser_models.py:188-457) -> f1tenth_gym_amd"""
from f1tenth_gym_amd.laser import ScanSimulator2D  # noqa: F401
from f1tenth_gym_amd.functional import check_ttc_jit, ray_cast  # noqa: F401
