"""Synthetic HalfCheetah-shaped environment (17-d obs, 6-d act), CPU oracle.

This is the benchmark environment of SURVEY.md section 8(d) cfg 2 -- it is NOT
in the reference (which steps real gym/MuJoCo envs); what it reproduces from
the reference is the *vector-env protocol* the collector relies on
(torchrl/env/vecenv.py:41-65): ``reset() -> (N, D)``, ``step(a) -> (obs (N, D),
rew (N, 1), done (N, 1) bool, {'time_limit': (N,) bool})``,
``partial_reset(mask (N,)) -> whole obs array``, ``seed(s)`` giving env ``i``
the seed ``s * N + i``.

Dynamics (fp32):  obs' = tanh(obs @ A + act @ B),  A (17x17), B (6x17) =
0.1 * RandomState(1234).randn;  reward = reward_scale * (obs'[0] - 0.1 |act|^2)
(the scale only in training mode, as torchrl/env/base_wrapper.py:31-40);
done = time_limit = (steps_in_episode >= horizon).  Reset observations are
N(0,1) from the Philox stream keyed (seed*N+i, episode_idx) (oracle/philox.py).
"""
import numpy as np
from . import philox

OBS_DIM = 17
ACT_DIM = 6


def dynamics_matrices(obs_dim=OBS_DIM, act_dim=ACT_DIM):
    rs = np.random.RandomState(1234)
    a = (0.1 * rs.randn(obs_dim, obs_dim)).astype(np.float32)
    b = (0.1 * rs.randn(act_dim, obs_dim)).astype(np.float32)
    return a, b


class _Box:
    """Minimal stand-in for gym.spaces.Box (shape/low/high only)."""
    def __init__(self, low, high, shape):
        self.low = np.full(shape, low, dtype=np.float32)
        self.high = np.full(shape, high, dtype=np.float32)
        self.shape = tuple(shape)


class SynthVecEnvCPU:
    """Vectorised numpy implementation (all N envs in one array op)."""

    def __init__(self, env_nums, horizon=1000, reward_scale=1.0,
                 env_index_offset=0, total_env_nums=None, obs_dim=OBS_DIM, act_dim=ACT_DIM):
        self.env_nums = env_nums
        self.horizon = horizon
        self._reward_scale = reward_scale
        self.training = True
        self.obs_dim, self.act_dim = int(obs_dim), int(act_dim)      # (other task shapes: same dynamics family)
        self.A, self.B = dynamics_matrices(self.obs_dim, self.act_dim)
        self.observation_space = _Box(-np.inf, np.inf, (self.obs_dim,))
        self.action_space = _Box(-1.0, 1.0, (self.act_dim,))
        self._offset = env_index_offset
        self._total = total_env_nums if total_env_nums is not None else env_nums
        self.seed(0)

    # -- protocol (torchrl/env/vecenv.py:28-65) --
    def train(self):
        self.training = True

    def eval(self):
        self.training = False

    def close(self):
        pass

    def seed(self, seed):
        idx = np.arange(self.env_nums, dtype=np.int64) + self._offset
        self.env_seed = np.int64(seed) * np.int64(self._total) + idx
        self.episode_idx = np.full(self.env_nums, -1, dtype=np.int64)
        self.t = np.zeros(self.env_nums, dtype=np.int64)

    def _fresh_obs(self, mask):
        self.episode_idx[mask] += 1
        self.t[mask] = 0
        return philox.normal_vector(self.obs_dim, self.episode_idx[mask], 0,
                                    philox.TAG_RESET, self.env_seed[mask])

    def reset(self):
        mask = np.ones(self.env_nums, dtype=bool)
        self._obs = self._fresh_obs(mask)
        return self._obs

    def partial_reset(self, index_mask):
        index_mask = np.asarray(index_mask).astype(bool)
        if index_mask.any():
            self._obs = self._obs.copy()
            self._obs[index_mask] = self._fresh_obs(index_mask)
        return self._obs

    def step(self, actions):
        actions = np.asarray(actions, dtype=np.float32).reshape(self.env_nums, self.act_dim)
        pre = self._obs.astype(np.float32) @ self.A + actions @ self.B
        nxt = np.tanh(pre.astype(np.float32)).astype(np.float32)
        rew = nxt[:, 0] - np.float32(0.1) * np.sum(actions * actions, axis=1, dtype=np.float32)
        if self.training:
            rew = rew * np.float32(self._reward_scale)
        self.t += 1
        done = self.t >= self.horizon
        self._obs = nxt
        return nxt, rew.astype(np.float32)[:, None], done[:, None], {"time_limit": done.copy()}


class SynthSingleEnvCPU:
    """One env, python-scalar style -- what SubProcVecEnv workers step
    (torchrl/env/subproc_vecenv.py:10-51).  Used by the CPU baseline only."""

    def __init__(self, env_seed, horizon=1000, reward_scale=1.0):
        self.A, self.B = dynamics_matrices()
        self.env_seed = np.int64(env_seed)
        self.horizon = horizon
        self._reward_scale = reward_scale
        self.training = True
        self.episode_idx = -1
        self.t = 0
        self.observation_space = _Box(-np.inf, np.inf, (OBS_DIM,))
        self.action_space = _Box(-1.0, 1.0, (ACT_DIM,))

    def train(self):
        self.training = True

    def eval(self):
        self.training = False

    def close(self):
        pass

    def reset(self):
        self.episode_idx += 1
        self.t = 0
        self._obs = philox.normal_vector(OBS_DIM, self.episode_idx, 0,
                                         philox.TAG_RESET, self.env_seed)
        return self._obs

    def step(self, action):
        action = np.asarray(action, dtype=np.float32)
        nxt = np.tanh(self._obs @ self.A + action @ self.B).astype(np.float32)
        rew = float(nxt[0] - np.float32(0.1) * np.sum(action * action, dtype=np.float32))
        if self.training:
            rew *= self._reward_scale
        self.t += 1
        done = self.t >= self.horizon
        self._obs = nxt
        return nxt, rew, done, {"time_limit": done}
