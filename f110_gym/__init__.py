"""`f110_gym` — import-level drop-in for the reference package of that name, backed by the MI355X
build (f1tenth_gym_amd).  Code written against the reference keeps its imports:

    import gym
    env = gym.make('f110_gym:f110-v0', map=..., map_ext='.png', num_agents=2)
    from f110_gym.envs.base_classes import Integrator

The reference registers the id at import time (gym/f110_gym/__init__.py:1-5); so does this, when
gym is importable.  Without gym, `f110_gym.make('f110-v0', **kwargs)` builds the same env."""
from f1tenth_gym_amd import __version__  # noqa: F401

ENTRY_POINT = 'f110_gym.envs:F110Env'

try:  # gym is optional (absent from the build image)
    from gym.envs.registration import register as _register
except Exception:  # noqa: BLE001
    _register = None

if _register is not None:
    try:
        _register(id='f110-v0', entry_point=ENTRY_POINT)
    except Exception:  # noqa: BLE001  (re-import under some gym versions: already registered)
        pass


def make(env_id='f110-v0', **kwargs):
    """gym.make('f110_gym:f110-v0', **kwargs) without gym"""
    if env_id.split(':')[-1] != 'f110-v0':
        raise ValueError("unknown environment id %r (this package provides 'f110-v0')" % (env_id,))
    from f110_gym.envs import F110Env
    return F110Env(**kwargs)
