"""f110_gym.envs.f110_env (reference: f110_env.py:53-418) -> f1tenth_gym_amd.env"""
from f1tenth_gym_amd.env import F110Env, F110VecEnv  # noqa: F401
from f1tenth_gym_amd.sharded import ShardedVecEnv  # noqa: F401
