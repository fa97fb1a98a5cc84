"""Minimal `gym` stand-in (no gym/gymnasium in the image, no network).

Only what the hot path's callers touch: ``gym.spaces.Box/Discrete`` for the
``isinstance(env.action_space, gym.spaces.Box)`` checks of the reference
(torchrl/algo/rl_algo.py:35, collector/base.py:26) and ``gym.make`` for the
synthetic ids.  Registered as ``sys.modules['gym']`` only when the real package
is absent.
"""
from . import spaces  # noqa: F401


class Env:
    pass


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self.action_space = getattr(env, "action_space", None)
        self.observation_space = getattr(env, "observation_space", None)

    def step(self, action):
        return self.env.step(action)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def seed(self, seed=None):
        return self.env.seed(seed)

    def close(self):
        return self.env.close()


def make(env_id, **kwargs):
    raise RuntimeError("gym is not installed; only the synthetic on-GPU env ids of "
                       "torchrl_amd.env (e.g. 'SynthHalfCheetah-v0') are available, via get_vec_env")
