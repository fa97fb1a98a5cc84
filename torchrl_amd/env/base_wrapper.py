"""Running observation normaliser for the on-GPU vector envs
(reference: torchrl/env/base_wrapper.py:44-121, applied by get_vec_env, env/get_env.py:69-77).

`Normalizer` keeps {mean, var, count} as one fp64 device buffer and runs the batch moments -> Chan
merge -> clip((x - mean) / (sqrt(var) + 1e-4)) chain as one kernel launch per vector step
(trl_norm_update_filt_f32).  With env shards on several GPUs the batch moments are all-reduced (SUM)
before the merge, so every rank holds the statistics one process would.

`NormObs` wraps a device vector env with the reference's protocol: `reset()` / `step()` return the
normalised observation (statistics are updated in training mode only); everything else -- including
`partial_reset` -- is forwarded to the wrapped env untouched, exactly like the reference's
`BaseWrapper.__getattr__` (its Q14: the array `partial_reset` returns is the RAW observation of ALL
envs, so the collector's next policy input is un-normalised after any reset).  That behaviour is
kept by default because the parity fixtures are generated from the reference;
`normalize_partial_reset=True` filters that array instead (no statistics update).
The collectors never call `env.step`: they read `env._obs_normalizer` and run the same kernels
inside their per-step launch sequence (torchrl_amd/collector/on_policy.py).
"""
import torch

from .. import _C
from .. import dist


class Normalizer:
    def __init__(self, shape, clip=10.0, device=None):
        self.shape = tuple(shape)
        if len(self.shape) != 1 or not 0 < self.shape[0] <= 64:
            raise _C.TrlError("device Normalizer handles flat observations of up to 64 features, got %r" % (shape,))
        self.clip = float(clip)
        self.should_estimate = True
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        D = self.shape[0]
        st = torch.zeros(2 * D + 1, dtype=torch.float64)
        st[D:2 * D] = 1.0
        st[2 * D] = 1e-4                                       # base_wrapper.py:64-68
        self.state = st.to(self.device)
        self._sums = torch.zeros(2 * D + 1, dtype=torch.float64, device=self.device)

    # ---- the reference's attribute surface (host copies; one small D2H each) ----
    @property
    def _mean(self):
        return self.state[:self.shape[0]].cpu().numpy()

    @property
    def _var(self):
        return self.state[self.shape[0]:2 * self.shape[0]].cpu().numpy()

    @property
    def _count(self):
        return float(self.state[-1].item())

    def stop_update_estimate(self):
        self.should_estimate = False

    def __getstate__(self):                                    # rl_algo.py:84-89 pickles the normaliser
        d = dict(self.__dict__)
        d["state"] = self.state.cpu()
        d["_sums"] = None
        d["device"] = str(self.device)
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self.device = torch.device(d["device"]) if torch.cuda.is_available() else torch.device("cpu")
        self.state = d["state"].to(self.device)
        self._sums = torch.zeros_like(self.state)

    def __deepcopy__(self, memo):                              # collector/base.py:129-130 deep-copies it for eval
        new = Normalizer(self.shape, self.clip, self.device)
        new.should_estimate = self.should_estimate
        new.state.copy_(self.state)
        return new

    # ---- kernels ----
    def _as_batch(self, data):
        x = torch.as_tensor(data).to(device=self.device, dtype=torch.float32)
        if x.dim() == 1:
            x = x.unsqueeze(0)
        return x.reshape(-1, self.shape[0]).contiguous()

    def update_filt(self, data, update=True, out=None):
        """update_estimate (when `update` and should_estimate) followed by filt, one launch."""
        x = self._as_batch(data)
        if out is None:
            out = torch.empty_like(x)
        update = bool(update and self.should_estimate)
        if update and dist.collectives_active():               # sharded envs: global batch statistics
            _C.norm_batch_moments(x, self._sums)
            dist.all_reduce_sum_(self._sums)
            _C.norm_merge(self.state, self._sums, self.shape[0])
            return _C.norm_filt(x, self.state, out, self.clip)
        return _C.norm_update_filt(x, self.state, out, self.clip, update)

    def update_estimate(self, data):
        if not self.should_estimate:
            return
        x = self._as_batch(data)
        if dist.collectives_active():
            _C.norm_batch_moments(x, self._sums)
            dist.all_reduce_sum_(self._sums)
            _C.norm_merge(self.state, self._sums, self.shape[0])
        else:
            _C.norm_update_filt(x, self.state, None, self.clip, True)

    def filt(self, raw):
        x = self._as_batch(raw)
        return _C.norm_filt(x, self.state, torch.empty_like(x), self.clip)

    filt_torch = filt

    def inverse_torch(self, raw):
        D = self.shape[0]
        return raw * torch.sqrt(self.state[D:2 * D]).to(raw.dtype) + self.state[:D].to(raw.dtype)

    inverse = inverse_torch


class NormObs:
    is_device_env = True

    def __init__(self, env, epsilon=1e-4, clipob=10.0, normalize_partial_reset=False):
        self._wrapped_env = env
        self.training = True
        self.clipob = clipob
        self.normalize_partial_reset = bool(normalize_partial_reset)
        self._obs_normalizer = Normalizer(env.observation_space.shape, clip=clipob, device=env.device)

    def __getattr__(self, attr):                               # base_wrapper.py:23-26
        if attr == "_wrapped_env":
            raise AttributeError()
        return getattr(self._wrapped_env, attr)

    def __setattr__(self, name, value):
        if name == "_reward_scale":                            # collector/base.py:35 sets it through the wrapper
            setattr(self._wrapped_env, name, value)
        else:
            object.__setattr__(self, name, value)

    def train(self):
        self._wrapped_env.train()
        self.training = True

    def eval(self):
        self._wrapped_env.eval()
        self.training = False

    def observation(self, observation):                        # base_wrapper.py:116-119
        return self._obs_normalizer.update_filt(observation, update=self.training)

    def reset(self, **kwargs):
        return self.observation(self._wrapped_env.reset(**kwargs))

    def step(self, action):
        obs, rew, done, info = self._wrapped_env.step(action)
        return self.observation(obs), rew, done, info

    def partial_reset(self, index_mask, **kwargs):
        raw = self._wrapped_env.partial_reset(index_mask, **kwargs)
        return self._obs_normalizer.filt(raw) if self.normalize_partial_reset else raw
