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

Also the SubProcVecEnv-style process-parallel stepping used by the CPU
baseline (torchrl/env/subproc_vecenv.py:10-51, 123-140).
"""
import multiprocessing as mp
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


# ---------------------------------------------------------------- CPU baseline
def _worker(env_fns, pipe):
    envs = [fn() for fn in env_fns]
    while True:
        cmd, data = pipe.recv()
        if cmd == "step":
            pipe.send([e.step(np.squeeze(a)) for e, a in zip(envs, data)])
        elif cmd == "reset":
            pipe.send([e.reset() for e in envs])
        elif cmd == "partial_reset":
            pipe.send([envs[i].reset() for i in np.argwhere(data == 1).reshape(-1)])
        elif cmd == "close":
            pipe.close()
            break


class SubProcVecEnvCPU:
    """P spawned workers x N/P python envs, pickled pipes, np.split / np.stack /
    dict-of-arrays merge per step -- the reference's CPU vector env
    (subproc_vecenv.py:59-140)."""

    def __init__(self, proc_nums, env_nums, env_fns, example_env):
        assert env_nums % proc_nums == 0
        self.env_nums, self.proc_nums = env_nums, proc_nums
        self.per = env_nums // proc_nums
        ctx = mp.get_context("spawn")
        self.pipes, self.procs = [], []
        for p in range(proc_nums):
            parent, child = ctx.Pipe()
            proc = ctx.Process(target=_worker,
                               args=(env_fns[p * self.per:(p + 1) * self.per], child),
                               daemon=True)
            proc.start()
            child.close()
            self.pipes.append(parent)
            self.procs.append(proc)
        self.observation_space = example_env.observation_space
        self.action_space = example_env.action_space
        self._reward_scale = 1

    def train(self):
        pass

    def eval(self):
        pass

    def reset(self):
        for p in self.pipes:
            p.send(("reset", None))
        obs = []
        for p in self.pipes:
            obs += p.recv()
        self._obs = np.stack(obs)
        return self._obs

    def partial_reset(self, mask):
        for m, p in zip(np.split(mask, self.proc_nums), self.pipes):
            p.send(("partial_reset", m))
        part = []
        for p in self.pipes:
            part += p.recv()
        self._obs[mask] = part
        return self._obs

    def step(self, actions):
        chunks = np.split(actions, self.env_nums)
        for i, p in enumerate(self.pipes):
            p.send(("step", chunks[i * self.per:(i + 1) * self.per]))
        res = []
        for p in self.pipes:
            res += p.recv()
        obs, rews, dones, infos = zip(*res)
        self._obs = np.stack(obs)
        merged = {k: np.array([i[k] for i in infos]) for k in infos[0]}
        return self._obs, np.stack(rews)[:, None], np.stack(dones)[:, None], merged

    def close(self):
        for p in self.pipes:
            try:
                p.send(("close", None))
            except Exception:
                pass
        for proc in self.procs:
            proc.join(timeout=5)
