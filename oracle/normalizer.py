"""Running observation normaliser, CPU oracle (test infrastructure only).

Restates torchrl/env/base_wrapper.py:
  * update_mean_var_count (:44-60) -- Chan's parallel merge of (mean, var, count) with a batch;
  * Normalizer (:63-95) -- fresh state mean 0, var 1, count 1e-4; update_estimate merges the batch
    mean / population variance / row count; filt = clip((x - mean) / (sqrt(var) + 1e-4), +-clip);
  * NormObs (:98-121) as applied to a vector env by get_vec_env (env/get_env.py:69-77): reset() and
    step() pass the observation through update (training mode only) + filt; every other attribute --
    including partial_reset -- is forwarded to the wrapped env untouched (base_wrapper.py:23-26), so the
    array returned by partial_reset is the RAW observation of ALL envs.
All arithmetic is numpy float64, as in the reference.
"""
import numpy as np


def update_mean_var_count(mean, var, count, batch_mean, batch_var, batch_count):
    delta = batch_mean - mean
    tot = count + batch_count
    new_mean = mean + delta * batch_count / tot
    m2 = var * count + batch_var * batch_count + np.square(delta) * count * batch_count / tot
    return new_mean, m2 / tot, tot


class NormalizerOracle:
    def __init__(self, shape, clip=10.0):
        self.shape = shape
        self._mean = np.zeros(shape)
        self._var = np.ones(shape)
        self._count = 1e-4
        self.clip = clip
        self.should_estimate = True

    def update_estimate(self, data):
        if not self.should_estimate:
            return
        data = np.asarray(data)
        self._mean, self._var, self._count = update_mean_var_count(
            self._mean, self._var, self._count, np.mean(data, axis=0), np.var(data, axis=0), data.shape[0])

    def filt(self, raw):
        return np.clip((np.asarray(raw) - self._mean) / (np.sqrt(self._var) + 1e-4), -self.clip, self.clip)

    def state(self):
        return np.concatenate([self._mean, self._var, [self._count]])


class NormObsOracle:
    """NormObs over a vector env with the reference's forwarding behaviour."""

    def __init__(self, env, clipob=10.0):
        self._wrapped_env = env
        self.training = True
        self._obs_normalizer = NormalizerOracle(env.observation_space.shape, clip=clipob)

    def __getattr__(self, attr):                       # base_wrapper.py:23-26 -- partial_reset lands here
        if attr == "_wrapped_env":
            raise AttributeError()
        return getattr(self._wrapped_env, attr)

    def train(self):
        self.training = True

    def eval(self):
        self.training = False

    def observation(self, observation):                # base_wrapper.py:116-119
        if self.training:
            self._obs_normalizer.update_estimate(observation)
        return self._obs_normalizer.filt(observation)

    def reset(self, **kwargs):
        return self.observation(self._wrapped_env.reset(**kwargs))

    def step(self, action):
        obs, rew, done, info = self._wrapped_env.step(action)
        return self.observation(obs), rew, done, info
