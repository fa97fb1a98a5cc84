"""Host-side vector env over ordinary Python envs (torchrl/env/vecenv.py:6-78) and its bridge to the device
collectors.

`VecEnv` keeps the reference's protocol -- `reset()` -> (N, D); `step(actions (N, A))` -> (obs (N, D),
rewards (N, 1), dones (N, 1) bool, infos: key -> array over the envs that reported it); `partial_reset(mask)`
returns the WHOLE observation array; `seed(s)` gives env i the seed `s * N + i`; unknown attributes fall through
to the first env -- for envs with the gym interface (`reset()`, `step(a) -> (obs, reward, done, info)`,
`observation_space`, `action_space`, optional `seed / train / eval / close / render`).

The physics of such envs runs on the host, one Python call per env and step: that part is the reference's cost
and stays it.  Everything else of a collector step -- policy and value networks, exploration noise, log-probs,
time-limit bootstrap, episode bookkeeping, the replay write -- runs in the same HIP kernels as for the on-GPU
envs; `HostEnvBridge` is the adaptor the collectors wrap a `VecEnv` in (two small host syncs per step: actions
out, observations / rewards / dones in)."""
import numpy as np
import torch


class VecEnv:
    is_device_env = False
    # The reference's partial_reset writes the fresh observations INTO the array `step` just returned (vecenv.py:50), and
    # its collectors add the sample after the reset (collector/base.py:203-227, on_policy.py:132-151): the `next_obs` row
    # they store for an env that was reset in a step -- by `done` or by max_episode_frames, where `terminals` stays
    # False and the TD target bootstraps from it -- is the RESET observation.  True (default): reproduce exactly that
    # (tests/golden/collect_hostenv.npz).  False: the stored transition keeps the observation the env produced.
    alias_reset_obs = True

    def __init__(self, env_nums, env_funcs, env_args):
        self.env_nums = int(env_nums)
        if isinstance(env_funcs, (list, tuple)):
            if len(env_funcs) != self.env_nums or len(env_args) != self.env_nums:
                raise ValueError("need one env constructor and one argument tuple per env")
            self.env_funcs, self.env_args = list(env_funcs), list(env_args)
        else:
            self.env_funcs = [env_funcs] * self.env_nums
            self.env_args = [env_args] * self.env_nums
        self.set_up_envs()

    def set_up_envs(self):
        self.envs = [fn(*arg) for fn, arg in zip(self.env_funcs, self.env_args)]

    def _each(self, name, *args):
        for env in self.envs:
            fn = getattr(env, name, None)
            if callable(fn):
                fn(*args)

    def train(self):
        self._each("train")

    def eval(self):
        self._each("eval")

    def close(self):
        self._each("close")

    def render(self):
        self._each("render")

    def reset(self, **kwargs):
        self._obs = np.stack([np.asarray(env.reset()) for env in self.envs])
        return self._obs

    def partial_reset(self, index_mask, **kwargs):
        index_mask = np.asarray(index_mask).reshape(-1).astype(bool)
        if not self.alias_reset_obs:
            self._obs = self._obs.copy()                                   # leave the array `step` returned alone
        for index in np.flatnonzero(index_mask):
            self._obs[index] = np.asarray(self.envs[index].reset())
        return self._obs

    def step(self, actions):
        actions = np.asarray(actions)
        results = [env.step(np.squeeze(a)) for env, a in zip(self.envs, np.split(actions, self.env_nums))]
        obs, rews, dones, infos = zip(*results)
        self._obs = np.stack([np.asarray(o) for o in obs])
        merged = {}
        for info in infos:                                   # merge_with(np.array, *infos)
            for key, value in (info or {}).items():
                merged.setdefault(key, []).append(value)
        merged = {key: np.array(values) for key, values in merged.items()}
        return (self._obs, np.stack(rews).astype(np.float64)[:, np.newaxis],
                np.stack(dones).astype(bool)[:, np.newaxis], merged)

    def seed(self, seed):
        for idx, env in enumerate(self.envs):
            if callable(getattr(env, "seed", None)):
                env.seed(seed * self.env_nums + idx)

    @property
    def observation_space(self):
        return self.envs[0].observation_space

    @property
    def action_space(self):
        return self.envs[0].action_space

    def __getattr__(self, attr):
        if attr in ("envs", "_wrapped_env"):
            raise AttributeError(attr)
        return getattr(self.envs[0], attr)


class HostEnvBridge:
    """What the device collectors need from an env, for a host `VecEnv`: device mirrors of the current observations
    and of the per-env step / return counters, and the two host round trips of a step."""
    is_device_env = False
    is_host_env = True
    kind = "vector"

    def __init__(self, venv, device):
        self.venv = venv
        self.env_nums = int(venv.env_nums)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("collectors run their networks on the GPU: pass device='cuda:<n>'")
        obs_shape, act_space = venv.observation_space.shape, venv.action_space
        self.discrete = hasattr(act_space, "n")                           # Discrete(n): actions are (N,) integers
        if len(obs_shape) != 1 or not (self.discrete or (hasattr(act_space, "shape") and len(act_space.shape) == 1)):
            raise ValueError("HostEnvBridge drives flat-observation envs with Box or Discrete actions "
                             "(got observation %r, action %r)" % (obs_shape, act_space))
        self.obs_dim = int(obs_shape[0])
        self.act_dim = 1 if self.discrete else int(act_space.shape[0])    # stored action width
        self.action_num = int(act_space.n) if self.discrete else 0
        n = self.env_nums
        self.cur_obs = torch.zeros(n, self.obs_dim, device=self.device)
        self.cur_step = torch.zeros(n, dtype=torch.int32, device=self.device)
        self.ep_return = torch.zeros(n, device=self.device)
        self.training = True
        # episode length if the envs advertise one (gym's TimeLimit, or a `horizon` attribute); 0: unknown
        self.horizon = int(getattr(venv, "_max_episode_steps", None) or getattr(venv, "horizon", None) or 0)

    # ---- the reference protocol, with device observations ----
    @property
    def observation_space(self):
        return self.venv.observation_space

    @property
    def action_space(self):
        return self.venv.action_space

    def train(self):
        self.training = True
        self.venv.train()

    def eval(self):
        self.training = False
        self.venv.eval()

    def close(self):
        self.venv.close()

    def render(self):
        self.venv.render()

    def seed(self, seed):
        self.venv.seed(seed)

    def _upload(self, array, out):
        out.copy_(torch.as_tensor(np.ascontiguousarray(array, dtype=np.float32)).view(out.shape), non_blocking=True)
        return out

    def reset(self, **kwargs):
        self._upload(self.venv.reset(**kwargs), self.cur_obs)
        self.cur_step.zero_()
        self.ep_return.zero_()
        return self.cur_obs

    # ---- the two host round trips of a collector step ----
    def host_step(self, act, next_obs, rewards, dones, time_limits=None):
        """act (N, A) device -> envs; fills the device rows next_obs (N, D), rewards, dones, time_limits (N, 1)."""
        a = act.detach().cpu().numpy()
        obs, rew, done, infos = self.venv.step(a.reshape(-1).astype(np.int64) if self.discrete else a.astype(np.float64))
        self._upload(obs, next_obs)
        self.cur_obs.copy_(next_obs)
        self._upload(rew, rewards)
        self._upload(done, dones)
        if time_limits is not None:
            if "time_limit" in infos and len(infos["time_limit"]) == self.env_nums:   # collector/base.py:213-215
                self._upload(infos["time_limit"], time_limits)
            else:
                time_limits.zero_()
        return done

    def host_partial_reset(self, mask, stored_next_obs=None):
        """mask (N,) uint8 device: reset those envs; `cur_obs` becomes the env's whole observation array.
        stored_next_obs: the ring row this step's `next_obs` went to when it holds env.step's own array (no observation
        wrapper in between) -- with `alias_reset_obs` it then reads what the reference stores, the array AFTER the reset."""
        m = mask.cpu().numpy().astype(bool)
        if m.any():
            self._upload(self.venv.partial_reset(m), self.cur_obs)
            if stored_next_obs is not None and getattr(self.venv, "alias_reset_obs", False):
                stored_next_obs.copy_(self.cur_obs)
        return m
