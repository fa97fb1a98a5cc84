"""On-GPU synthetic vector environments.

`SynthVecEnv` keeps the reference's vector-env protocol
(torchrl/env/vecenv.py:6-78: env_nums, reset, partial_reset(mask) returning the
whole obs array, seed(s) -> env i seeded s*N+i, train/eval/close,
observation_space / action_space, settable _reward_scale) but all state lives
in device tensors: on-policy collection steps it inside the fused rollout kernel
(trl_rollout_synth_f32), `step()` / the off-policy collector use the stand-alone step kernel
(trl_synth_env_step_f32) -- there is no per-env Python object and no host copy.

HalfCheetah-shaped dynamics (17-d obs, 6-d act):
    obs' = tanh(obs @ A + act @ B),  A (17x17), B (6x17) = 0.1 * RandomState(1234).randn
    reward = reward_scale * (obs'[0] - 0.1 |act|^2)   (scale only in train mode)
    done = time_limit = (steps since reset >= horizon)
    reset obs ~ N(0,1), Philox stream keyed (seed*N_total + global_index, episode_idx)
`index_offset` / `total_env_nums` place a shard of a larger logical vector env
on this GPU (multi-GPU: envs are sharded by index).
"""
import numpy as np
import torch
from gym import spaces

from .. import _C

SYNTH_IDS = {"SynthHalfCheetah-v0": dict(obs_dim=17, act_dim=6, horizon=1000)}
SYNTH_FRAME_IDS = {"SynthAtari-v0": dict(frame_shape=(4, 84, 84), action_num=6, horizon=1000)}


def dynamics_matrices(obs_dim, act_dim):
    rs = np.random.RandomState(1234)
    a = (0.1 * rs.randn(obs_dim, obs_dim)).astype(np.float32)
    b = (0.1 * rs.randn(act_dim, obs_dim)).astype(np.float32)
    return a, b


class SynthVecEnv:
    is_device_env = True

    def __init__(self, env_nums, obs_dim=17, act_dim=6, horizon=1000, reward_scale=1.0,
                 device=None, index_offset=0, total_env_nums=None):
        self.env_nums = int(env_nums)
        self.obs_dim, self.act_dim, self.horizon = obs_dim, act_dim, int(horizon)
        self._reward_scale = reward_scale
        self.training = True
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.index_offset = int(index_offset)
        self.total_env_nums = int(total_env_nums) if total_env_nums is not None else self.env_nums
        self.observation_space = spaces.Box(-np.inf, np.inf, (obs_dim,))
        self.action_space = spaces.Box(-1.0, 1.0, (act_dim,))
        a, b = dynamics_matrices(obs_dim, act_dim)
        self.env_A = torch.from_numpy(a).to(self.device)
        self.env_B = torch.from_numpy(b).to(self.device)
        n = self.env_nums
        self.cur_obs = torch.zeros(n, obs_dim, device=self.device)
        self.t_env = torch.zeros(n, dtype=torch.int32, device=self.device)
        self.cur_step = torch.zeros(n, dtype=torch.int32, device=self.device)   # collector-side counter
        self.episode_idx = torch.full((n,), -1, dtype=torch.int32, device=self.device)
        self.ep_return = torch.zeros(n, device=self.device)
        self.seed(0)

    # ---- protocol ----
    def train(self):
        self.training = True

    def eval(self):
        self.training = False

    def close(self):
        pass

    def render(self):
        raise NotImplementedError("synthetic envs have nothing to render")

    def seed(self, seed):
        self._seed = int(seed)
        self.seed_base = self._seed * self.total_env_nums + self.index_offset
        self.episode_idx.fill_(-1)

    @property
    def effective_reward_scale(self):
        return float(self._reward_scale) if self.training else 1.0

    def reset(self, **kwargs):
        _C.synth_reset(self.cur_obs, self.t_env, self.cur_step, self.episode_idx, self.ep_return, None,
                       self.seed_base)
        return self.cur_obs

    def partial_reset(self, index_mask, **kwargs):
        mask = torch.as_tensor(index_mask).to(self.device).to(torch.uint8).contiguous()
        _C.synth_reset(self.cur_obs, self.t_env, self.cur_step, self.episode_idx, self.ep_return, mask,
                       self.seed_base)
        return self.cur_obs

    def step(self, actions):
        """One vector step as a stand-alone kernel (torchrl/env/vecenv.py:53-61 protocol, device tensors):
        returns (obs (N, D), rewards (N, 1), dones (N, 1) bool, {'time_limit': (N,) bool})."""
        n = self.env_nums
        acts = torch.as_tensor(actions).to(device=self.device, dtype=torch.float32).reshape(n, self.act_dim).contiguous()
        nxt = torch.empty(n, self.obs_dim, device=self.device)
        rew = torch.empty(n, 1, device=self.device)
        done = torch.empty(n, 1, device=self.device)
        _C.synth_env_step(self.cur_obs, acts, self.env_A, self.env_B, self.t_env, self.effective_reward_scale,
                          self.horizon, nxt, rew, done)
        dones = done > 0.5
        return nxt, rew, dones, {"time_limit": dones.reshape(n)}


class SynthFrameVecEnv:
    """Atari-shaped synthetic vector env (SURVEY.md section 8(d) cfg 5): uint8 frame stacks
    (N, 4, 84, 84) resident on the GPU, Discrete(6) actions.  Each step shifts the stack and appends
    one pseudo-random frame (Philox keyed by env seed and env time, torchrl_amd/csrc/k_dqn.hip);
    reward = 1 when the action equals (first byte of the new frame) % A; done = time_limit =
    (steps since reset >= horizon).  Same vector-env protocol as SynthVecEnv; frames are what the
    reference's WarpFrame + FrameStack wrappers would deliver, kept as bytes (the /255 - 0.5 of
    ScaledFloatFrame is applied inside the first conv layer's im2col)."""
    is_device_env = True
    kind = "frames"

    def __init__(self, env_nums, frame_shape=(4, 84, 84), action_num=6, horizon=1000, reward_scale=1.0,
                 device=None, index_offset=0, total_env_nums=None):
        self.env_nums = int(env_nums)
        self.frame_shape, self.action_num, self.horizon = tuple(frame_shape), int(action_num), int(horizon)
        self._reward_scale = reward_scale
        self.training = True
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.index_offset = int(index_offset)
        self.total_env_nums = int(total_env_nums) if total_env_nums is not None else self.env_nums
        self.observation_space = spaces.Box(0, 255, self.frame_shape, dtype=np.uint8)
        self.action_space = spaces.Discrete(self.action_num)
        n = self.env_nums
        self.cur_obs = torch.zeros((n,) + self.frame_shape, dtype=torch.uint8, device=self.device)
        self.t_env = torch.zeros(n, dtype=torch.int32, device=self.device)
        self.cur_step = torch.zeros(n, dtype=torch.int32, device=self.device)
        self.episode_idx = torch.zeros(n, dtype=torch.int32, device=self.device)
        self.ep_return = torch.zeros(n, device=self.device)
        self.seed(0)

    def train(self):
        self.training = True

    def eval(self):
        self.training = False

    def close(self):
        pass

    def seed(self, seed):
        self._seed = int(seed)
        self.seed_base = self._seed * self.total_env_nums + self.index_offset

    @property
    def effective_reward_scale(self):
        return float(self._reward_scale) if self.training else 1.0

    def reset(self, **kwargs):
        _C.synth_frames_reset(self.cur_obs, self.t_env, self.seed_base, None)
        self.cur_step.zero_()
        self.ep_return.zero_()
        return self.cur_obs

    def partial_reset(self, index_mask, **kwargs):
        mask = torch.as_tensor(index_mask).to(self.device).to(torch.uint8).contiguous()
        _C.synth_frames_reset(self.cur_obs, self.t_env, self.seed_base, mask)
        return self.cur_obs

    def step(self, actions):
        n = self.env_nums
        acts = torch.as_tensor(actions).to(device=self.device).reshape(n).to(torch.int64).contiguous()
        nxt = torch.empty_like(self.cur_obs)
        rew = torch.empty(n, 1, device=self.device)
        done = torch.empty(n, 1, device=self.device)
        _C.synth_frames_step(self.cur_obs, acts, self.t_env, self.seed_base, self.horizon, self.action_num, nxt, rew, done)
        dones = done > 0.5
        return nxt, rew, dones, {"time_limit": dones.reshape(n)}
