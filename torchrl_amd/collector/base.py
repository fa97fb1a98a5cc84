"""Collector bases (torchrl/collector/base.py:10-280).

The reference's collectors drive a Python loop: policy forward on the device,
`.cpu().numpy()`, `env.step`, numpy ring write -- per step.  Here the loop body
lives in one HIP kernel (torchrl_amd/csrc/k_rollout.hip) and a collector is the
host object that owns the launch descriptor: which networks, which env state,
which ring rows.  Constructor signatures, attribute names and the return
protocol (`train_one_epoch() -> {'train_rewards', 'train_epoch_reward'}`,
`eval_one_epoch() -> {'eval_rewards', 'eval_traj_length'}`) follow the
reference.
"""
import copy

import gym
import numpy as np
import torch

from .. import _C


class BaseCollector:
    def __init__(self, env, eval_env, pf, replay_buffer, epoch_frames, train_render=False,
                 eval_episodes=1, eval_render=False, device='cpu', max_episode_frames=999):
        self.pf = pf
        self.replay_buffer = replay_buffer
        self.env = env
        self.env.train()
        self.continuous = isinstance(self.env.action_space, gym.spaces.Box)
        self.train_render = train_render
        if eval_env is not None:
            self.eval_env = eval_env
        else:
            self.eval_env = copy.deepcopy(env)
        self.eval_env._reward_scale = 1
        self.eval_episodes = eval_episodes
        self.eval_render = eval_render
        self.device = torch.device(device)
        self.to(self.device)
        self.current_ob = self.env.reset()
        self.train_rew = 0
        self.epoch_frames = epoch_frames
        self.sample_epoch_frames = epoch_frames
        self.max_episode_frames = max_episode_frames
        self.current_step = 0
        self.train_rews = []

    def start_episode(self):
        pass

    def finish_episode(self):
        pass

    def terminate(self):
        self.env.close()
        self.eval_env.close()

    def to(self, device):
        for net in self.funcs.values():
            net.to(device)

    @property
    def funcs(self):
        return {"pf": self.pf}

    def train_one_epoch(self):
        self.train_rews = []
        self.train_epoch_reward = 0
        self.env.train()
        for _ in range(self.sample_epoch_frames):
            self.train_epoch_reward += self.take_actions()
        return {'train_rewards': self.train_rews, 'train_epoch_reward': self.train_epoch_reward}

    def take_actions(self):
        raise NotImplementedError("single-env host collectors are outside the GPU hot path; "
                                  "use VecOnPolicyCollector with a device env")


class VecCollector(BaseCollector):
    """Vector collector base: `epoch_frames // env_nums` vector steps per epoch
    (torchrl/collector/base.py:176-182)."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.sample_epoch_frames //= self.env.env_nums
        if not getattr(self.env, "is_device_env", False):
            raise _C.TrlError("torchrl_amd collectors need an on-GPU env (torchrl_amd.env.get_vec_env); "
                              "host gym envs have no kernel path")
