"""Vectorised on-policy rollout step, CPU oracle.

Restates torchrl/collector/on_policy.py:84-155 (VecOnPolicyCollector) on top of
torchrl/collector/base.py:176-230 (VecCollector: ``epoch_frames // N`` steps
per epoch, per-env step counters and running returns):

  act = tanh(mean(obs) + std * eps), eps ~ CPU torch generator (Q5);
  value = vf(obs); env.step; episode-return bookkeeping (on_policy.py:126-130);
  if any done or any over-length: V' = vf(next_obs) for all envs,
  terminals = done | surpass, rewards += discount * V' * surpass,
  partial_reset(done | surpass), counters of those envs -> 0 (:132-148);
  one ring-buffer row per step (:151).

The SubProcVecEnv-style process-parallel stepping used by the CPU baseline
lives in oracle/subproc_env.py (numpy only, so spawned workers start fast).
"""
import numpy as np
import torch
from . import nets


class VecOnPolicyCollectorOracle:
    def __init__(self, env, ring, pf_params, logstd, vf_params, epoch_frames,
                 max_episode_frames=999, discount=0.99, act="tanh", tanh_action=True):
        self.env, self.ring = env, ring
        self.pf, self.logstd, self.vf = pf_params, logstd, vf_params
        self.act, self.tanh_action = act, tanh_action
        self.discount = discount
        self.max_episode_frames = max_episode_frames
        n = env.env_nums
        self.steps_per_epoch = epoch_frames // n           # base.py:179
        self.current_step = np.zeros((n, 1))
        self.train_rew = np.zeros((n, 1))
        self.env.train()
        self.current_ob = env.reset()

    @torch.no_grad()
    def take_actions(self, noise=None):
        ob = torch.as_tensor(np.asarray(self.current_ob), dtype=torch.float32)
        if noise is None:                                   # distribution.py:67-70
            noise = torch.randn(ob.shape[0], self.logstd.numel())
        acts = nets.explore_action(ob, self.pf, self.logstd, noise, self.act,
                                   self.tanh_action).numpy()
        values = nets.mlp(ob, self.vf, self.act).numpy()
        next_obs, rewards, dones, infos = self.env.step(acts)
        self.current_step += 1
        sample = {"obs": self.current_ob, "next_obs": next_obs, "acts": acts,
                  "values": values, "rewards": rewards, "terminals": dones,
                  "time_limits": infos["time_limit"][:, None]
                  if "time_limit" in infos else [False]}
        self.train_rew += rewards
        if np.any(dones):
            self.train_rews += list(self.train_rew[dones])
            self.train_rew[dones] = 0
        surpass = self.current_step >= self.max_episode_frames
        if np.any(dones) or np.any(surpass):
            last_v = nets.mlp(torch.as_tensor(next_obs, dtype=torch.float32),
                              self.vf, self.act).numpy()
            flag = dones | surpass
            sample["terminals"] = flag
            sample["rewards"] = rewards + self.discount * last_v * surpass
            next_obs = self.env.partial_reset(np.squeeze(flag, axis=-1))
            self.current_step[flag] = 0
        self.ring.add(sample)
        self.current_ob = next_obs
        return np.sum(rewards)

    def train_one_epoch(self, noise=None):
        self.train_rews = []
        total = 0
        for t in range(self.steps_per_epoch):
            total += self.take_actions(None if noise is None else noise[t])
        return {"train_rewards": self.train_rews, "train_epoch_reward": total}


class VecCollectorOracle:
    """Off-policy vector collector, restating torchrl/collector/base.py:176-230: explore (one
    N(0,1) draw per step, no log-probs), env.step, running returns logged and cleared on done,
    reset of envs with done | step >= max_episode_frames, one ring row per step holding the RAW
    rewards / dones (no bootstrap on the off-policy path)."""

    def __init__(self, env, ring, pf_params, epoch_frames, max_episode_frames=999, act="relu", tanh_action=True):
        from .sac import rsample
        self._rsample = rsample
        self.env, self.ring, self.pf = env, ring, pf_params
        self.act, self.tanh_action = act, tanh_action
        self.max_episode_frames = max_episode_frames
        n = env.env_nums
        self.steps_per_epoch = epoch_frames // n
        self.current_step = np.zeros((n, 1))
        self.train_rew = np.zeros((n, 1))
        self.env.train()
        self.current_ob = env.reset()

    @torch.no_grad()
    def take_actions(self, noise=None):
        ob = torch.as_tensor(np.asarray(self.current_ob), dtype=torch.float32)
        head = nets.mlp(ob, self.pf, self.act)
        if noise is None:
            noise = torch.randn(ob.shape[0], head.shape[1] // 2)
        act = self._rsample(head, torch.as_tensor(noise, dtype=torch.float32), self.tanh_action)[0].numpy()
        next_ob, reward, done, infos = self.env.step(act)
        self.current_step += 1
        sample = {"obs": self.current_ob, "next_obs": next_ob, "acts": act, "rewards": reward, "terminals": done,
                  "time_limits": infos["time_limit"][:, None] if "time_limit" in infos else [False]}
        self.train_rew += reward
        if np.any(done):
            self.train_rews += list(self.train_rew[done])
            self.train_rew[done] = 0
        flag = (self.current_step >= self.max_episode_frames) | done
        if np.any(flag):
            next_ob = self.env.partial_reset(np.squeeze(flag, axis=-1))
            self.current_step[flag] = 0
        self.ring.add(sample)
        self.current_ob = next_ob
        return np.sum(reward)

    def train_one_epoch(self, noise=None):
        self.train_rews = []
        total = 0
        for t in range(self.steps_per_epoch):
            total += self.take_actions(None if noise is None else noise[t])
        return {"train_rewards": self.train_rews, "train_epoch_reward": total}


def eval_one_epoch(eval_env, act_fn, eval_episodes=1):
    """Greedy evaluation, restating torchrl/collector/base.py:232-280: per round reset every env, step until each env
    has finished once (finished envs are reset and keep stepping, their later rewards masked out), collect the return
    and length of every env's FIRST episode.  `act_fn(obs ndarray) -> actions ndarray` is the policy's eval_act."""
    eval_env.eval()
    rews_all, lens_all = [], []
    n = eval_env.env_nums
    for _ in range(eval_episodes):
        epi_done = np.zeros((n, 1), dtype=bool)
        obs = eval_env.reset()
        rews = np.zeros((n, 1))
        traj_len = np.zeros((n, 1))
        while not np.all(epi_done):
            obs, r, done, _ = eval_env.step(act_fn(np.asarray(obs)))
            rews = rews + (1 - epi_done) * r
            traj_len = traj_len + (1 - epi_done)
            epi_done = epi_done | done
            if np.any(done):
                obs = eval_env.partial_reset(np.squeeze(done, axis=-1))
        rews_all += list(rews)
        lens_all += list(traj_len)
    return {"eval_rewards": rews_all, "eval_traj_length": float(np.mean(lens_all))}

