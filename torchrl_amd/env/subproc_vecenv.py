"""`SubProcVecEnv`: the host `VecEnv` protocol with the envs stepped in worker processes
(torchrl/env/subproc_vecenv.py:10-157) -- `proc_nums` spawned workers, each owning `env_nums / proc_nums`
consecutive envs, one pipe round trip per worker and call.  Env constructors and their arguments must be picklable
(module-level callables).  `seed(s)` reaches the envs here (env i gets s * env_nums + i, like `VecEnv`); in the
reference the workers drop the command (its Q15)."""
import multiprocessing as mp

import numpy as np

from .vecenv import VecEnv


def _worker(env_funcs, env_args, pipe):
    envs = [fn(*arg) for fn, arg in zip(env_funcs, env_args)]

    def each(name, *a):
        for env in envs:
            fn = getattr(env, name, None)
            if callable(fn):
                fn(*a)
    try:
        while True:
            command, data = pipe.recv()
            if command == "step":
                pipe.send([env.step(np.squeeze(a)) for env, a in zip(envs, data)])
            elif command == "reset":
                pipe.send([np.asarray(env.reset(**data)) for env in envs])
            elif command == "partial_reset":
                mask, kwargs = data
                pipe.send([np.asarray(envs[i].reset(**kwargs)) for i in np.flatnonzero(mask)])
            elif command == "seed":
                for env, s in zip(envs, data):
                    if callable(getattr(env, "seed", None)):
                        env.seed(s)
            elif command in ("train", "eval", "render"):
                each(command)
            elif command == "close":
                break
    except (EOFError, KeyboardInterrupt):
        pass
    finally:
        each("close")
        pipe.close()


class SubProcVecEnv(VecEnv):
    def __init__(self, proc_nums, env_nums, env_funcs, env_args):
        self.proc_nums = int(proc_nums)
        super().__init__(env_nums, env_funcs, env_args)

    def set_up_envs(self):
        if self.proc_nums <= 0 or self.env_nums % self.proc_nums != 0:
            raise ValueError("env_nums (%d) must be a multiple of proc_nums (%d)" % (self.env_nums, self.proc_nums))
        self.example_env = self.env_funcs[0](*self.env_args[0])          # spaces and attribute look-ups
        self.env_nums_per_proc = self.env_nums // self.proc_nums
        ctx = mp.get_context("spawn")
        self.workers, self.pipes = [], []
        for i in range(self.proc_nums):
            lo, hi = i * self.env_nums_per_proc, (i + 1) * self.env_nums_per_proc
            parent, child = ctx.Pipe()
            p = ctx.Process(target=_worker, args=(self.env_funcs[lo:hi], self.env_args[lo:hi], child), daemon=True)
            p.start()
            child.close()
            self.workers.append(p)
            self.pipes.append(parent)
        self._closed = False

    def _send_all(self, command, per_worker):
        for pipe, data in zip(self.pipes, per_worker):
            pipe.send((command, data))

    def _gather(self):
        out = []
        for pipe in self.pipes:
            out += pipe.recv()
        return out

    def train(self):
        self._send_all("train", [None] * self.proc_nums)

    def eval(self):
        self._send_all("eval", [None] * self.proc_nums)

    def render(self):
        self._send_all("render", [None] * self.proc_nums)

    def close(self):
        if getattr(self, "_closed", True):
            return
        self._closed = True
        self._send_all("close", [None] * self.proc_nums)
        for p in self.workers:
            p.join(timeout=5)
        for pipe in self.pipes:
            pipe.close()
        if callable(getattr(self.example_env, "close", None)):
            self.example_env.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, **kwargs):
        self._send_all("reset", [kwargs] * self.proc_nums)
        self._obs = np.stack(self._gather())
        return self._obs

    def partial_reset(self, index_mask, **kwargs):
        index_mask = np.asarray(index_mask).reshape(-1).astype(bool)
        self._send_all("partial_reset", [(m, kwargs) for m in np.split(index_mask, self.proc_nums)])
        fresh = self._gather()
        if not self.alias_reset_obs:                                   # see VecEnv.alias_reset_obs
            self._obs = self._obs.copy()
        for index, ob in zip(np.flatnonzero(index_mask), fresh):
            self._obs[index] = ob
        return self._obs

    def step(self, actions):
        per_env = np.split(np.asarray(actions), self.env_nums)
        k = self.env_nums_per_proc
        self._send_all("step", [per_env[i * k:(i + 1) * k] for i in range(self.proc_nums)])
        obs, rews, dones, infos = zip(*self._gather())
        self._obs = np.stack([np.asarray(o) for o in obs])
        merged = {}
        for info in infos:
            for key, value in (info or {}).items():
                merged.setdefault(key, []).append(value)
        merged = {key: np.array(values) for key, values in merged.items()}
        return (self._obs, np.stack(rews).astype(np.float64)[:, np.newaxis],
                np.stack(dones).astype(bool)[:, np.newaxis], merged)

    def seed(self, seed):
        k = self.env_nums_per_proc
        self._send_all("seed", [[seed * self.env_nums + i * k + j for j in range(k)] for i in range(self.proc_nums)])

    @property
    def observation_space(self):
        return self.example_env.observation_space

    @property
    def action_space(self):
        return self.example_env.action_space

    def __getattr__(self, attr):
        if attr in ("example_env", "_wrapped_env", "envs"):
            raise AttributeError(attr)
        return getattr(self.example_env, attr)

    def __deepcopy__(self, memo):
        """A second set of workers on fresh envs (collectors deep-copy the env for evaluation when none is given)."""
        return SubProcVecEnv(self.proc_nums, self.env_nums, self.env_funcs, self.env_args)
