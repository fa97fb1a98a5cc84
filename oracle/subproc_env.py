"""SubProcVecEnv-style process-parallel env stepping, CPU oracle / baseline.

Restates torchrl/env/subproc_vecenv.py:10-51, 59-140: P spawned workers, each
owning N/P per-env Python objects, one multiprocessing.Pipe per worker, pickled
lists of per-env actions / (obs, rew, done, info) tuples per step, np.split /
np.stack / dict-of-arrays merge in the parent.  numpy only (no torch import) so
spawned workers start quickly.
"""
import multiprocessing as mp

import numpy as np

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
