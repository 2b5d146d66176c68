"""The names `f110_gym.envs` exports in the reference (envs/__init__.py:1-5 star-imports its four
modules), bound to the MI355X implementations."""
from f110_gym.envs.f110_env import F110Env, F110VecEnv, ShardedVecEnv  # noqa: F401
from f110_gym.envs.dynamic_models import *  # noqa: F401,F403
from f110_gym.envs.laser_models import *  # noqa: F401,F403
from f110_gym.envs.base_classes import Integrator, RaceCar, Simulator  # noqa: F401
from f110_gym.envs.collision_models import *  # noqa: F401,F403
