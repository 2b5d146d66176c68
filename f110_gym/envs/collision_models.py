"""f110_gym.envs.collision_models (reference: collision_models.py:113-260) -> f1tenth_gym_amd.functional"""
from f1tenth_gym_amd.functional import get_vertices, collision, collision_multiple  # noqa: F401
