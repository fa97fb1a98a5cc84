"""VecOnPolicyCollector on the fused rollout kernel
(reference: torchrl/collector/on_policy.py:84-155).

`train_one_epoch()` is ONE kernel launch for all `epoch_frames // N` steps: the
launch descriptor points the kernel at the policy / value parameter blocks, the
env's device state and the ring rows `[top, top + T)` of the replay buffer; the
kernel does act -> value -> env step -> over-length bootstrap -> partial reset
-> row write for every step (see k_rollout.hip).  `take_actions()` is the same
launch with one step.

Exploration noise:
  * noise_mode="host" (default, reference parity): one `torch.randn(N, A)` per
    step from the CPU generator -- the reference's own stream (its Q5) --
    uploaded as a (T, N, A) tensor;
  * noise_mode="device": Philox4x32-10 in the kernel keyed by (env seed, global
    step) -- statistically equivalent, no host work; what bench.py uses.
"""
import numpy as np
import torch

from .. import _C
from .base import VecCollector, BaseCollector


class OnPolicyCollectorBase(BaseCollector):
    def __init__(self, vf, discount=0.99, **kwargs):
        self.vf = vf
        super().__init__(**kwargs)
        self.discount = discount

    @property
    def funcs(self):
        return {"pf": self.pf, "vf": self.vf}


class VecOnPolicyCollector(VecCollector):
    def __init__(self, vf, discount=0.99, noise_mode="host", **kwargs):
        self.vf = vf
        super().__init__(noise_mode=noise_mode, **kwargs)
        self.discount = discount
        self._check_shapes()

    @property
    def funcs(self):
        return {"pf": self.pf, "vf": self.vf}

    def _check_shapes(self):
        ps, vs = self.pf.mlp2_spec(), self.vf.mlp2_spec()
        if ps is None or vs is None:
            raise _C.TrlError("fused collector needs MLP2 policy/value nets (two equal hidden layers, "
                              "Tanh or ReLU, no LayerNorm)")
        if ps[0] != self.env.obs_dim or ps[2] != self.env.act_dim or vs[:2] != ps[:2] or vs[2] != 1 or vs[3] != ps[3]:
            raise _C.TrlError("policy %s / value %s shapes do not match env (%d obs, %d act)"
                              % (ps, vs, self.env.obs_dim, self.env.act_dim))
        if not hasattr(self.pf, "logstd"):
            raise _C.TrlError("fused collector supports GuassianContPolicyBasicBias policies")
        self._spec = ps

    # ---- launch ----
    def _launch(self, env, n_steps, store, deterministic, noise, max_frames=None):
        D, H, A, act = self._spec
        buf = self.replay_buffer
        a = _C.RolloutArgs()
        a.pf_params = self.pf.flat_params().data_ptr()
        a.vf_params = self.vf.flat_params().data_ptr()
        a.D, a.H, a.A, a.act = D, H, A, act
        a.tanh_action = int(bool(self.pf.tanh_action))
        a.env_A, a.env_B = env.env_A.data_ptr(), env.env_B.data_ptr()
        a.reward_scale, a.horizon, a.env_seed_base = env.effective_reward_scale, env.horizon, env.seed_base
        a.cur_obs, a.t_env, a.cur_step = env.cur_obs.data_ptr(), env.t_env.data_ptr(), env.cur_step.data_ptr()
        a.episode_idx, a.ep_return = env.episode_idx.data_ptr(), env.ep_return.data_ptr()
        a.noise = noise.data_ptr() if noise is not None else None
        a.noise_step0 = self.global_step
        a.deterministic = int(deterministic)
        N = env.env_nums
        if store:
            feats = (("obs", D), ("next_obs", D), ("acts", A), ("values", 1), ("rewards", 1),
                     ("terminals", 1), ("time_limits", 1), ("old_logp", 1))
            for key, f in feats:
                setattr(a, key, buf._ensure_key(key, (N, f)).data_ptr())
            a.rows, a.top = buf._max_replay_buffer_size, buf._top
        else:
            a.rows, a.top = 1, 0
        a.N, a.n_steps = N, n_steps
        a.max_episode_frames = int(self.max_episode_frames if max_frames is None else max_frames)
        a.discount = float(self.discount)
        a.epoch_reward, a.ep_count, a.ep_log = (self._epoch_reward.data_ptr(), self._ep_count.data_ptr(),
                                                self._ep_log.data_ptr())
        a.ep_cap, a.step0 = self.EP_LOG_CAP, 0
        self._clear_header()
        _C.rollout(a, env.device)
        if store:
            buf._advance(n_steps)
            # log pi_old written by the kernel covers the whole ring only for a full-ring launch
            buf._old_logp_fresh = (n_steps == buf._max_replay_buffer_size)

    def _host_noise(self, n_steps, env):
        A = self._spec[2]
        draws = [torch.randn(env.env_nums, A) for _ in range(n_steps)]    # the reference's stream, step by step
        return torch.stack(draws).to(env.device, non_blocking=True).contiguous()

    def rollout(self, n_steps):
        """Enqueue `n_steps` vector steps into the replay buffer; no host sync."""
        self.env.train()
        noise = self._host_noise(n_steps, self.env) if self.noise_mode == "host" else None
        self._launch(self.env, n_steps, True, False, noise)
        self.global_step += n_steps
        self.current_ob = self.env.cur_obs

    def train_one_epoch(self):
        self.rollout(self.sample_epoch_frames)
        self.train_epoch_reward, cnt = self._read_header()                # one 16-byte D2H per epoch ...
        log = self._finished_episodes(cnt)                                 # ... plus the episode log when any ended
        self.train_rews = [np.float32(r) for r in log[:, 2]]
        return {'train_rewards': self.train_rews, 'train_epoch_reward': self.train_epoch_reward}

    def take_actions(self):
        self.rollout(1)
        return float(self._epoch_reward.item())

    def eval_one_epoch(self):
        """Greedy evaluation (torchrl/collector/base.py:232-280): every eval env plays its
        first episode with action = tanh(mean); nothing is written to the replay buffer."""
        env = self.eval_env
        env.eval()
        rews, lens = [], []
        for _ in range(self.eval_episodes):
            env.reset()
            self._launch(env, env.horizon, False, True, None, max_frames=2 ** 31 - 1)
            log = self._finished_episodes()
            first = {}
            for step, idx, ret in log:
                first.setdefault(int(idx), (ret, int(step) + 1))
            rews += [np.float32(first[i][0]) for i in sorted(first)]
            lens += [first[i][1] for i in sorted(first)]
        return {"eval_rewards": rews, "eval_traj_length": float(np.mean(lens)) if lens else 0.0}
