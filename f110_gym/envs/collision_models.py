"""f110_gym.envs.collision_models (reference: collision_models.py:34-260) -> f1tenth_gym_amd.functional"""
from f1tenth_gym_amd.functional import (perpendicular, tripleProduct, avgPoint, indexOfFurthestPoint, support, collision,  # noqa: F401
                                        collision_multiple, get_trmtx, get_vertices)
