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
from collections.abc import Mapping

import gym
import os

import numpy as np
import torch

from .. import _C


class _CollectorBase:
    def __init__(self, env, eval_env, pf, replay_buffer, epoch_frames, train_render=False,
                 eval_episodes=1, eval_render=False, device='cpu', max_episode_frames=999):
        from ..env.vecenv import HostEnvBridge, VecEnv
        if isinstance(env, VecEnv):                                          # host Python envs: bridge to the device path
            env = HostEnvBridge(env, device)
        if eval_env is None and getattr(env, "is_host_env", False) and not hasattr(env, "_obs_normalizer"):
            eval_env = HostEnvBridge(copy.deepcopy(env.venv), device)
        if isinstance(eval_env, VecEnv):
            eval_env = HostEnvBridge(eval_env, device)
        self.pf = pf
        self.replay_buffer = replay_buffer
        self.env = env
        self.env.train()
        space = self.env.action_space                                       # Box of this package's shim or of a real gym
        self.continuous = isinstance(space, gym.spaces.Box) or (hasattr(space, "shape") and not hasattr(space, "n"))
        self.train_render = train_render
        if eval_env is not None:
            self.eval_env = eval_env
        else:
            self.eval_env = copy.deepcopy(env)
        self.eval_env._reward_scale = 1
        if hasattr(env, "_obs_normalizer"):                                 # collector/base.py:33-34
            self.eval_env._obs_normalizer = env._obs_normalizer
        self.eval_episodes = eval_episodes
        self.eval_render = eval_render
        self.device = torch.device(device)
        env_dev = getattr(env, "device", None)
        if env_dev is not None and self.device.type == "cuda" and torch.device(env_dev).type == "cuda":
            want = self.device.index if self.device.index is not None else torch.cuda.current_device()
            have = torch.device(env_dev).index
            if have is not None and have != want:                              # kernels take raw pointers: no implicit peer access
                raise _C.TrlError("collector device cuda:%d but the env lives on cuda:%d -- call torch.cuda.set_device() "
                                  "before building envs / buffers, or pass device= to get_vec_env" % (want, have))
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


class _EpochResult(Mapping):
    """train_one_epoch's result, read back when first looked at (see VecCollector.train_one_epoch)."""

    def __init__(self, col, event):
        self._col, self._event, self._data = col, event, None

    def resolve(self):
        if self._data is None:
            col = self._col
            self._event.synchronize()
            h = col._hdr_host
            col.train_epoch_reward, cnt = float(h[0]), int(h[1:].view(torch.int32)[0])
            col._rendezvous_check()
            if cnt <= col.SPECULATIVE_ROWS:                                 # everything is already on the host
                log = col._ep_log_host[:cnt].numpy().copy()
                log = log[np.lexsort((log[:, 1], log[:, 0]))] if cnt else np.zeros((0, 3), dtype=np.float32)
            else:
                log = col._finished_episodes(cnt)
            col.train_rews = list(log[:, 2])
            self._data = {'train_rewards': col.train_rews, 'train_epoch_reward': col.train_epoch_reward}
            if col._pending is self:
                col._pending = None
        return self._data

    def __getitem__(self, key): return self.resolve()[key]
    def __iter__(self): return iter(self.resolve())
    def __len__(self): return len(self.resolve())


class VecCollector(_CollectorBase):
    """Off-policy vector collector (torchrl/collector/base.py:176-280) on the device env.

    One vector step = policy MLP on the MFMA layer kernels -> reparameterised TanhNormal sample ->
    stand-alone env step writing next_obs / rewards / terminals straight into the replay row ->
    bookkeeping kernel (step counters, running returns, reset mask = done | step >= max frames) ->
    partial reset.  `epoch_frames // env_nums` vector steps per epoch (base.py:179); nothing is read
    back until the epoch ends.  Exploration noise: CPU `torch.randn(N, A)` per step (reference
    stream, its Q5) or the device Philox stream (`noise_mode="device"`).
    """
    EP_LOG_CAP = 1 << 16

    def __init__(self, noise_mode="host", eval_env=None, **kwargs):
        super().__init__(eval_env=eval_env, **kwargs)
        self.sample_epoch_frames //= self.env.env_nums
        if not (getattr(self.env, "is_device_env", False) or getattr(self.env, "is_host_env", False)):
            raise _C.TrlError("torchrl_amd collectors drive an on-GPU env (torchrl_amd.env.get_vec_env) or a host "
                              "torchrl_amd.env.VecEnv of Python envs")
        if noise_mode not in ("host", "device"):
            raise ValueError("noise_mode must be 'host' or 'device'")
        self.noise_mode = noise_mode
        self.eager_epoch_result = False         # True: train_one_epoch waits for its result instead of handing back a lazy mapping
        self.global_step = 0
        self._log_step0 = 0
        dev = self.env.device
        # Epoch reward (f64) and finished-episode count (i32) share one 16-byte header; TWO such headers (the fused rollout
        # alternates between them and zeroes the idle one inside its launch: no memset launch per epoch) and the episode
        # log sit in ONE allocation with a page-locked twin, so that header + the head of the log come back in one D2H.
        self._blob = torch.zeros(8 + 3 * self.EP_LOG_CAP, device=dev)
        self._hdr2 = self._blob[:8].view(torch.float64).view(2, 2)
        self._ep_log = self._blob[8:].view(self.EP_LOG_CAP, 3)
        self._blob_host = torch.zeros(8 + 3 * self.EP_LOG_CAP).pin_memory() if torch.cuda.is_available() else None
        self._ep_log_host = None if self._blob_host is None else self._blob_host[8:].view(self.EP_LOG_CAP, 3)
        self._idle_hdr_clean = False            # the header not in use is known to be zero (a fused rollout cleared it)
        self._use_header(0)
        if self._blob_host is not None:
            # the runtime sets up its device-to-host copy path for a size class on first use (milliseconds): pay that
            # here, not in the first epoch in which episodes end
            for n in (64, 4096, self.EP_LOG_CAP):
                self._blob_host[:8 + 3 * n].copy_(self._blob[:8 + 3 * n], non_blocking=True)
            torch.cuda.current_stream(dev).synchronize()
        self._mask = torch.zeros(self.env.env_nums, dtype=torch.uint8, device=dev)
        self._noise_seed = 0xC011

    # ---- pieces shared with the on-policy subclass ----
    def _env_advance(self, env, act, nxt, rew, done, time_limits=None):
        """One env.step: fills the device rows next_obs / rewards / dones (/ time_limits)."""
        if getattr(env, "is_host_env", False):                              # host Python envs: actions out, results in
            env.host_step(act, nxt, rew, done, time_limits)
            return
        _C.synth_env_step(env.cur_obs, act, env.env_A, env.env_B, env.t_env, env.effective_reward_scale,
                          env.horizon, nxt, rew, done)
        if time_limits is not None:
            time_limits.copy_(done)                                         # synthetic env: time_limit == done

    def _env_reset_masked(self, env, stored_next_obs=None):
        """env.partial_reset(self._mask); env.cur_obs then holds the whole (raw) observation array.  stored_next_obs: the
        ring row holding env.step's own return of this step (host envs: see VecEnv.alias_reset_obs)."""
        if getattr(env, "is_host_env", False):
            env.host_partial_reset(self._mask, stored_next_obs)
        else:
            _C.synth_reset(env.cur_obs, env.t_env, env.cur_step, env.episode_idx, env.ep_return, self._mask, env.seed_base)

    def _use_header(self, i):
        self._hdr_i = i
        self._hdr = self._hdr2[i]
        self._epoch_reward = self._hdr[:1]
        self._ep_count = self._hdr[1:].view(torch.int32)[:1]
        self._hdr_host = None if self._blob_host is None else self._blob_host[4 * i:4 * i + 4].view(torch.float64)

    def _read_header(self):
        """(epoch reward, finished-episode count) with a single host sync (asynchronous copy into page-locked memory,
        then one wait on the stream)."""
        self._hdr_host.copy_(self._hdr, non_blocking=True)
        torch.cuda.current_stream(self._hdr.device).synchronize()
        h = self._hdr_host
        return float(h[0]), int(h[1:].view(torch.int32)[0])

    def _clear_header(self, swap=False, index=None):
        """Zero {epoch reward, episode count} for the launches that follow.  swap=True (the fused rollout, which clears
        the idle header inside its launch): switch to the idle header when it is known to be clean instead of launching a
        memset, and return the header the coming launch shall clear.  index: use THAT header (a captured launch sequence
        carries the header it was captured with)."""
        nxt = None
        if index is not None and index != self._hdr_i:
            self._use_header(index)
            self._idle_hdr_clean = False
        if swap and self._idle_hdr_clean:
            self._use_header(1 - self._hdr_i)
        else:
            self._hdr.zero_()
        if swap:
            nxt = self._hdr2[1 - self._hdr_i]
        self._idle_hdr_clean = False            # (the caller sets it once the clearing launch is enqueued)
        self._log_step0 = self.global_step      # the device log keeps steps RELATIVE to here (float32 columns: exact to 2^24)
        return nxt

    def _finished_episodes(self, cnt=None):
        """(step, env, return) rows of episodes that ended since the log was cleared, in the
        reference's list order (step-major, then env index)."""
        if cnt is None:
            cnt = int(self._ep_count.item())
        if cnt > self.EP_LOG_CAP and not getattr(self, "_ep_cap_warned", False):
            self._ep_cap_warned = True
            import logging
            logging.getLogger("torchrl_amd").warning("%d episodes ended in one epoch but the device log holds %d: "
                                                     "train_rewards is truncated", cnt, self.EP_LOG_CAP)
        cnt = min(cnt, self.EP_LOG_CAP)
        if not cnt:
            return np.zeros((0, 3), dtype=np.float32)
        self._ep_log_host[:cnt].copy_(self._ep_log[:cnt], non_blocking=True)
        torch.cuda.current_stream(self._ep_log.device).synchronize()
        log = self._ep_log_host[:cnt].numpy().copy()
        return log[np.lexsort((log[:, 1], log[:, 0]))]

    def _explore_noise(self, env):
        """(N, A) standard normals of this vector step: the CPU generator's draw (distribution.py:67-70) or the device
        Philox stream; with env shards on several ranks, this rank's rows of the draw for ALL envs."""
        from .. import dist
        n, a_dim = env.env_nums, env.act_dim
        if self.noise_mode == "host":
            make = lambda m, f: torch.randn(m, f)
        else:
            make = lambda m, f: _C.philox_normal(torch.empty(m, f, device=env.device), self._noise_seed, self.global_step)
        return dist.shard_rows_of_global(make, 1, n, a_dim, env.device).to(env.device, non_blocking=True)

    def _policy_action(self, env, deterministic, ob=None):
        """`ob`: what the policy sees -- env.cur_obs, or the normalised observation of a NormObs env."""
        from .. import ops
        pf = self.pf
        ob = env.cur_obs if ob is None else ob
        if not self.continuous:                                             # epsilon-greedy over a Q network
            if deterministic:
                return torch.as_tensor(pf.eval_act(ob)).to(env.device).reshape(-1).contiguous()
            return pf.explore(ob)["action"].reshape(-1).contiguous()
        if hasattr(pf, "norm_std_explore") or type(pf).__name__ == "DetContPolicy":
            # deterministic policies (continuous_policy.py:28-74): [tanh](mlp(obs)) (+ N(0, norm_std_explore))
            last = _C.ACT_TANH if pf.tanh_action else _C.ACT_NONE
            act, _ = ops.mlp_forward(ops.linear_layers(pf), ob, ops.act_code(pf), last_act=last, keep=False)
            sigma = float(getattr(pf, "norm_std_explore", 0.0))
            if deterministic or not sigma:
                return act
            return _C.noisy_action(act, self._explore_noise(env), sigma)
        if not hasattr(pf, "tanh_action") or hasattr(pf, "logstd"):
            raise _C.TrlError("VecCollector's kernel path expects a GuassianContPolicy (mean | log_std head); "
                              "state-independent-std policies use VecOnPolicyCollector")
        n, a_dim = env.env_nums, env.act_dim
        head, _ = ops.mlp_forward(ops.linear_layers(pf), ob, ops.act_code(pf), keep=False)
        eps = torch.zeros(n, a_dim, device=env.device) if deterministic else self._explore_noise(env)
        act, _ = _C.rsample_fwd(head, eps, bool(pf.tanh_action))
        return act

    def _step(self, env, store, deterministic=False, max_frames=None, ob=None):
        """One vector step.  On a NormObs env (`env._obs_normalizer`) `ob` is the policy input -- the normalised
        observation, or the RAW one of all envs right after any reset, the reference's Q14 -- and the next policy
        input is returned: obs / next_obs rows hold what the reference stores (env.step's normalised return,
        base_wrapper.py:116-121), the statistics are updated in training mode only."""
        buf = self.replay_buffer
        if getattr(env, "kind", "vector") == "frames":
            return self._step_frames(env, store, deterministic, max_frames)
        n, d, a_dim = env.env_nums, env.obs_dim, env.act_dim
        nz = getattr(env, "_obs_normalizer", None)
        if nz is not None and ob is None:
            raise _C.TrlError("VecCollector._step: a normalised env needs the policy input `ob`")
        pol_in = env.cur_obs if nz is None else ob
        if self._one_launch_step(env, nz):
            return self._step_one_launch(env, store, deterministic, max_frames)
        act = self._policy_action(env, deterministic, pol_in)
        if store:
            row = buf._top
            buf._ensure_key("obs", (n, d))[row].copy_(pol_in)                # before the env advances in place
            buf._ensure_key("acts", (n, a_dim))[row].copy_(act if self.continuous else act.reshape(n, 1))
            nxt = buf._ensure_key("next_obs", (n, d))[row]
            rew = buf._ensure_key("rewards", (n, 1))[row]
            done = buf._ensure_key("terminals", (n, 1))[row]
        else:
            nxt = torch.empty(n, d, device=env.device)
            rew = torch.empty(n, 1, device=env.device)
            done = torch.empty(n, 1, device=env.device)
        raw_next = nxt if nz is None else torch.empty(n, d, device=env.device)
        self._env_advance(env, act, raw_next, rew, done, buf._ensure_key("time_limits", (n, 1))[row] if store else None)
        if nz is not None:
            nz.update_filt(raw_next, update=env.training, out=nxt)           # NormObs.observation on env.step's return
        _C.collector_bookkeep(rew, done, env.cur_step, env.ep_return,
                              self.max_episode_frames if max_frames is None else max_frames, self._mask,
                              self._epoch_reward, self._ep_count, self._ep_log, self.global_step - self._log_step0)
        self._env_reset_masked(env, nxt if (store and nz is None) else None)
        if store:
            buf._advance()
        self.global_step += 1
        if nz is None:
            return env.cur_obs
        # partial_reset bypasses the wrapper and returns the whole RAW array (base_wrapper.py:23-26, vecenv.py:47-51)
        alt = nz.filt(env.cur_obs) if getattr(env, "normalize_partial_reset", False) else env.cur_obs
        return _C.select_on_mask(self._mask, alt, nxt, torch.empty(n, d, device=env.device))

    def _one_launch_step(self, env, nz):
        """Synthetic vector env + reparameterised Gaussian policy, no normaliser: everything after the policy MLP is
        ONE launch (trl_synth_collect_step_f32) instead of eight."""
        pf = self.pf
        return (nz is None and self.continuous and not getattr(env, "is_host_env", False)
                and hasattr(pf, "tanh_action") and not hasattr(pf, "logstd") and not hasattr(pf, "norm_std_explore")
                and type(pf).__name__ != "DetContPolicy" and hasattr(env, "env_A"))

    def _step_one_launch(self, env, store, deterministic, max_frames):
        from .. import ops
        buf, pf = self.replay_buffer, self.pf
        n, d, a_dim = env.env_nums, env.obs_dim, env.act_dim
        head, _ = ops.mlp_forward(ops.linear_layers(pf), env.cur_obs, ops.act_code(pf), keep=False)
        noise = None
        if deterministic:
            if getattr(self, "_zero_eps", None) is None or self._zero_eps.shape[0] != n:
                self._zero_eps = torch.zeros(n, a_dim, device=env.device)
            eps = self._zero_eps
        elif self.noise_mode == "device":                                # drawn inside the launch: this rank's rows of the
            from .. import dist                                          # (all envs, A) Philox draw of this vector step
            eps, noise = None, (self._noise_seed, self.global_step, dist.rank() * n)
        else:
            eps = self._explore_noise(env)
        if store:
            row = buf._top
            rows = (buf._ensure_key("obs", (n, d))[row], buf._ensure_key("acts", (n, a_dim))[row],
                    buf._ensure_key("next_obs", (n, d))[row], buf._ensure_key("rewards", (n, 1))[row],
                    buf._ensure_key("terminals", (n, 1))[row], buf._ensure_key("time_limits", (n, 1))[row])
        else:
            f = lambda *shape: torch.empty(shape, device=env.device)
            rows = (None, None, f(n, d), f(n, 1), f(n, 1), None)
        _C.synth_collect_step(env, head, eps, env.cur_step, env.ep_return,
                              self.max_episode_frames if max_frames is None else max_frames, rows, self._mask,
                              self._epoch_reward, self._ep_count, self._ep_log, self.global_step - self._log_step0,
                              bool(pf.tanh_action), noise=noise)
        if store:
            buf._advance()
        self.global_step += 1
        return env.cur_obs

    def _step_frames(self, env, store, deterministic, max_frames):
        """Discrete-action step on the uint8 frame env: frames stay bytes in the replay rows; actions are
        stored as (N, 1) so that DQN's gather works (the reference squeezes them, its Q16)."""
        buf = self.replay_buffer
        n, shape = env.env_nums, env.frame_shape
        act = self._policy_action(env, deterministic)                       # (N,) int64
        dedup = store and hasattr(buf, "append_step")                    # frame-deduplicating replay
        if dedup:
            if buf._stream is None:
                buf.begin_episodes(env.cur_obs)                             # stacks of the running episodes
            row = buf._top
            buf.mark_obs_row()
            buf._ensure_key("acts", (n, 1))[row].copy_(act.unsqueeze(-1))
            nxt = None
            rew = buf._ensure_key("rewards", (n, 1))[row]
            done = buf._ensure_key("terminals", (n, 1))[row]
        elif store:
            row = buf._top
            buf._ensure_key("obs", (n,) + shape, dtype=torch.uint8)[row].copy_(env.cur_obs)
            buf._ensure_key("acts", (n, 1))[row].copy_(act.unsqueeze(-1))
            nxt = buf._ensure_key("next_obs", (n,) + shape, dtype=torch.uint8)[row]
            rew = buf._ensure_key("rewards", (n, 1))[row]
            done = buf._ensure_key("terminals", (n, 1))[row]
        else:
            nxt = None
            rew = torch.empty(n, 1, device=env.device)
            done = torch.empty(n, 1, device=env.device)
        _C.synth_frames_step(env.cur_obs, act, env.t_env, env.seed_base, env.horizon, env.action_num, nxt, rew, done)
        if dedup:
            buf.append_step(env.cur_obs)                                    # the one new frame of next_obs
        if store:
            buf._ensure_key("time_limits", (n, 1))[row].copy_(done)
        _C.collector_bookkeep(rew, done, env.cur_step, env.ep_return,
                              self.max_episode_frames if max_frames is None else max_frames, self._mask,
                              self._epoch_reward, self._ep_count, self._ep_log, self.global_step - self._log_step0)
        _C.synth_frames_reset(env.cur_obs, env.t_env, env.seed_base, self._mask)
        if dedup:
            buf.begin_episodes(env.cur_obs, self._mask)                     # fresh stacks of the envs just reset
        if store:
            buf._advance()
        self.global_step += 1

    def _capture_key_extras(self, env, net):
        """Everything else a captured collection sequence bakes in besides the ring: the env's seed base, the bookkeeping
        tensors (step / return counters, reset mask, header, episode log, env clock) and the parameter storages of the
        network it runs -- a re-seed, a `.to()` or a checkpoint load that replaces tensors must force a fresh eager pass and
        a re-capture instead of replaying stale values or freed pointers."""
        ptr = lambda t: t.data_ptr() if isinstance(t, torch.Tensor) else None
        params = [p for p in net.parameters()] if net is not None else []
        return (int(getattr(env, "seed_base", 0)), int(getattr(env, "action_num", 0) or 0), ptr(getattr(env, "t_env", None)),
                ptr(getattr(env, "cur_step", None)), ptr(getattr(env, "ep_return", None)), ptr(getattr(env, "episode_idx", None)),
                ptr(self._mask), ptr(self._hdr), ptr(self._ep_log), id(net), len(params),
                params[0].data_ptr() if params else None, params[-1].data_ptr() if params else None)

    def _replayed_rollout(self, n_steps):
        """Training collection on the synthetic vector env with device noise, one rank: a vector step = the policy pass +
        trl_synth_collect_step_f32 with its device-side state, whose step counter / ring row / epoch start live on the device -- captured into
        a HIP graph (all `n_steps` of the call) on the second call and replayed afterwards: one host call per epoch.
        Returns False when the configuration is outside that path."""
        from .. import dist, ops
        env, buf, pf = self.env, self.replay_buffer, self.pf
        if not (self._one_launch_step(env, getattr(env, "_obs_normalizer", None)) and self.noise_mode == "device"
                and dist.world_size() == 1 and os.environ.get("TRL_NO_GRAPH") != "1"
                and hasattr(buf, "_obs") and buf._max_replay_buffer_size > 0):
            return False
        n, d, a_dim = env.env_nums, env.obs_dim, env.act_dim
        ring = [buf._ensure_key("obs", (n, d)), buf._ensure_key("acts", (n, a_dim)), buf._ensure_key("next_obs", (n, d)),
                buf._ensure_key("rewards", (n, 1)), buf._ensure_key("terminals", (n, 1)), buf._ensure_key("time_limits", (n, 1))]
        if getattr(self, "_dyn", None) is None:
            self._dyn = torch.zeros(4, dtype=torch.int64, device=env.device)
            self._dyn_stager = _C.PinnedStager(4, torch.int64)              # (the host may be epochs ahead of the device)
            self._step_graph, self._step_key, self._step_seen = None, None, None
        host = self._dyn_stager.stage()
        host[0], host[1], host[2], host[3] = self.global_step, buf._top, self._log_step0, 0
        self._dyn_stager.upload(self._dyn)                                 # one 32-byte upload per epoch
        key = tuple(t.data_ptr() for t in ring) + (env.cur_obs.data_ptr(), n, d, a_dim, int(self.max_episode_frames),
                                                   bool(pf.tanh_action), int(env.horizon), n_steps) \
            + self._capture_key_extras(env, pf)

        def one_step():
            head, _ = ops.mlp_forward(ops.linear_layers(pf), env.cur_obs, ops.act_code(pf), keep=False)
            _C.synth_collect_step_dyn(env, head, env.cur_step, env.ep_return, self.max_episode_frames, ring, self._dyn,
                                      self._mask, self._epoch_reward, self._ep_count, self._ep_log, bool(pf.tanh_action),
                                      self._noise_seed, 0)
        # all n_steps of the call are ONE graph (the per-step state is on the device): one host call per epoch
        if self._step_graph is not None and self._step_key == key:
            self._step_graph.replay()
        elif self._step_seen != key:
            self._step_seen, self._step_graph = key, None                  # first visit: eager (warm-up)
            for _ in range(n_steps):
                one_step()
        else:
            graph, _ = _C.capture_graph(lambda: [one_step() for _ in range(n_steps)])
            self._step_graph, self._step_key = graph, key
            graph.replay()
        buf._advance(n_steps)
        self.global_step += n_steps
        return True

    def _replayed_rollout_frames(self, n_steps):
        """Training collection on the uint8 frame env with an epsilon-greedy Q network, one rank, plain ring: the launches
        of all `n_steps` vector steps (conv forward, action, frame step, bookkeeping, masked reset) are ONE replayed HIP
        graph.  What the host decides per step goes in ahead of the replay -- the numpy draws of every step in the reference's
        order (discrete_policies.py:58-65: rand, then randint, per step) with the epsilon comparison already applied (the
        kernel gets u in {0, 1} against a fixed 0.5), one upload each -- and the ring position lives on the device: the frame
        step files the whole transition into row `ring_row[0]` itself (trl_synth_frames_collect_u8: the pre-step stacks are
        stored while they are shifted), the reset launch that ends a step advances the row, one 8-byte upload per epoch sets
        it.  Same ring contents, header, episode log and env state as the step-by-step path (tests/test_dqn_gpu.py).
        Returns False when the configuration is outside that path."""
        from .. import dist
        env, buf, pf = self.env, self.replay_buffer, self.pf
        if not (getattr(env, "kind", "vector") == "frames" and not self.continuous and hasattr(pf, "decay_frames")
                and hasattr(pf, "quantile_num") and dist.world_size() == 1 and os.environ.get("TRL_NO_GRAPH") != "1"
                and not hasattr(buf, "append_step") and hasattr(buf, "_ensure_key") and buf._max_replay_buffer_size > 0
                and not getattr(env, "is_host_env", False)):
            return False
        n, shape, dev = env.env_nums, tuple(env.frame_shape), env.device
        ring = (buf._ensure_key("obs", (n,) + shape, dtype=torch.uint8), buf._ensure_key("next_obs", (n,) + shape, dtype=torch.uint8),
                buf._ensure_key("acts", (n, 1)), buf._ensure_key("rewards", (n, 1)), buf._ensure_key("terminals", (n, 1)),
                buf._ensure_key("time_limits", (n, 1)))
        rows = int(ring[0].shape[0])
        A, Q = int(pf.action_shape), int(pf.quantile_num)
        key = tuple(t.data_ptr() for t in ring) + (env.cur_obs.data_ptr(), n, shape, A, Q, int(self.max_episode_frames),
                                                   int(env.horizon), n_steps, id(pf.qf), rows) \
            + self._capture_key_extras(env, pf.qf)
        st = getattr(self, "_fr", None)
        if st is None or st["key"] != key:
            st = self._fr = {"key": key, "graph": None, "seen": False,
                             "rew": torch.empty(n, 1, device=dev), "done": torch.empty(n, 1, device=dev),
                             "row": torch.zeros(1, dtype=torch.int64, device=dev), "row_stager": _C.PinnedStager(1, torch.int64),
                             "u": torch.empty(n_steps, n, device=dev), "ra": torch.empty(n_steps, n, dtype=torch.int64, device=dev),
                             "u_stager": _C.PinnedStager(n_steps * n, torch.float32),
                             "ra_stager": _C.PinnedStager(n_steps * n, torch.int64)}
        # the host's part of every step, in the reference's order
        u_host = st["u_stager"].stage().view(n_steps, n).numpy()
        ra_host = st["ra_stager"].stage().view(n_steps, n).numpy()
        for t in range(n_steps):
            pf.count += 1
            if pf.count < pf.decay_frames:
                pf.epsilon = pf.start_epsilon - (pf.start_epsilon - pf.end_epsilon) * (pf.count / pf.decay_frames)
            else:
                pf.epsilon = pf.end_epsilon
            u = np.random.rand(n, 1).astype(np.float32).reshape(-1)
            ra_host[t] = np.random.randint(low=0, high=A, size=(n, 1)).reshape(-1)
            u_host[t] = np.where(u < np.float32(pf.epsilon), np.float32(0.0), np.float32(1.0))
        st["u_stager"].upload(st["u"].view(-1))
        st["ra_stager"].upload(st["ra"].view(-1))
        st["row_stager"].stage()[0] = buf._top                            # (the host may be epochs ahead of the device)
        st["row_stager"].upload(st["row"])

        def one_step(t):
            # behind the Q network: the action launch (head + epsilon-greedy for narrow heads, trl_dqn_act_f32; it advances
            # the ring row from the second step on), then ONE launch for the frame step, the filing of the transition, the
            # collector's bookkeeping and the reset of the envs that ended
            act = pf.act_on(env.cur_obs, st["u"][t], st["ra"][t], 0.5, want_q=False,
                            ring_row=st["row"] if t > 0 else None, n_rows=rows)[1]
            _C.synth_frames_collect(env.cur_obs, act, env.t_env, env.seed_base, env.horizon, env.action_num, ring, st["row"],
                                    st["rew"], st["done"],
                                    book=(env.cur_step, env.ep_return, self.max_episode_frames, self._mask,
                                          self._epoch_reward, self._ep_count, self._ep_log, t))

        if self.global_step != self._log_step0:
            raise _C.TrlError("frame rollout replay: the episode log was not cleared for this rollout")
        hdr_key = self._hdr_i                                              # (a captured sequence carries its header)
        if st["graph"] is not None and st.get("hdr") == hdr_key:
            st["graph"].replay()
        elif not st["seen"] or st["graph"] is not None:
            st["seen"], st["graph"] = True, None                           # first visit (or another header): eager
            for t in range(n_steps):
                one_step(t)
        else:
            st["graph"], _ = _C.capture_graph(lambda: [one_step(t) for t in range(n_steps)])
            st["hdr"] = hdr_key
            st["graph"].replay()
        buf._advance(n_steps)
        self.global_step += n_steps
        return True

    def rollout(self, n_steps):
        """Enqueue `n_steps` vector steps into the replay buffer; no host sync."""
        self._resolve_pending()                                            # the header / episode log are about to be reused
        self.env.train()
        self._clear_header()
        if self._replayed_rollout(n_steps) or self._replayed_rollout_frames(n_steps):
            self.current_ob = self.env.cur_obs
            return
        ob = None
        if hasattr(self.env, "_obs_normalizer"):
            ob = torch.as_tensor(self.current_ob).to(device=self.env.device, dtype=torch.float32).contiguous()
        for _ in range(n_steps):
            ob = self._step(self.env, True, ob=ob)
        self.current_ob = self.env.cur_obs if ob is None else ob

    SPECULATIVE_ROWS = 4096             # episode-log rows copied along with the header before their count is known

    def train_one_epoch(self):
        """Returns the reference's {'train_rewards', 'train_epoch_reward'} (base.py:60-68 / on_policy.py:277-286) as a
        mapping that is read back on FIRST ACCESS: the header and the head of the episode log are copied to page-locked
        memory behind the rollout in stream order, so a caller that launches the update before looking at the result
        (RLAlgo.train does) never leaves the GPU idle for the read-back.  The next rollout resolves a result nobody looked
        at (`collector.eager_epoch_result = True`: read back before returning)."""
        self._resolve_pending()
        self._published = False                                            # only THIS epoch's fused launch may set it (a per-step
        self.rollout(self.sample_epoch_frames)                             # rollout must not inherit a stale "already published")
        if self.eager_epoch_result or self._ep_log_host is None \
                or getattr(self.env, "is_host_env", False):
            return self._epoch_result_now()
        k = self.SPECULATIVE_ROWS                                          # both headers + the head of the log: ONE copy
        if not getattr(self, "_published", False):                         # (the fused rollout's value pass wrote them already)
            self._blob_host[:8 + 3 * k].copy_(self._blob[:8 + 3 * k], non_blocking=True)
        self._published = False
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(self._hdr.device))
        self._pending = _EpochResult(self, done)
        return self._pending

    def _resolve_pending(self):
        pending = getattr(self, "_pending", None)
        if pending is not None:
            pending.resolve()

    def _epoch_result_now(self):
        self.train_epoch_reward, cnt = self._read_header()                # one 16-byte D2H per epoch ...
        self._rendezvous_check()
        log = self._finished_episodes(cnt)                                 # ... plus the episode log when any ended
        self.train_rews = list(log[:, 2])                            # np.float32 scalars
        return {'train_rewards': self.train_rews, 'train_epoch_reward': self.train_epoch_reward}

    def _rendezvous_check(self):
        pass                                                               # (the on-policy rollout kernel has one)

    def take_actions(self):
        self.rollout(1)
        return float(self._epoch_reward.item())

    def _eval_steps(self, env):
        """Device envs end every episode at `horizon`; host envs run until each env has reported `done` once
        (`while not np.all(epi_done)`, base.py:252) -- at most `horizon` steps when they advertise one."""
        if getattr(env, "is_host_env", False):
            return int(env.horizon) if env.horizon else 1 << 30
        return int(env.horizon)

    def eval_one_epoch(self):
        """Greedy evaluation (base.py:232-280): action = tanh(mean); first episode of every eval env."""
        self._resolve_pending()                                            # (the header / episode log are reused below)
        env = self.eval_env
        if hasattr(self.env, "_obs_normalizer"):                           # collector/base.py:236-237
            env._obs_normalizer = copy.deepcopy(self.env._obs_normalizer)
        env.eval()
        rews, lens = [], []
        for _ in range(self.eval_episodes):
            ob = env.reset()
            self._clear_header()
            for _ in range(self._eval_steps(env)):
                ob = self._step(env, False, deterministic=True, max_frames=2 ** 31 - 1, ob=ob)
                if getattr(env, "is_host_env", False) and int(self._ep_count.item()) >= env.env_nums \
                        and len({int(i) for _, i, _ in self._finished_episodes()}) == env.env_nums:
                    break                                                   # every env has finished its first episode
            first = {}
            for step, idx, ret in self._finished_episodes():
                first.setdefault(int(idx), (ret, int(step) + 1))   # steps are logged relative to the clear
            rews += [np.float32(first[i][0]) for i in sorted(first)]
            lens += [first[i][1] for i in sorted(first)]
        return {"eval_rewards": rews, "eval_traj_length": float(np.mean(lens)) if lens else 0.0}


class BaseCollector(VecCollector):
    """The reference's single-env off-policy collector (torchrl/collector/base.py:10-174): here the vector collector
    on a one-env env (`torchrl.env.get_env`), so the single-env example scripts run on the same kernels."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        if self.env.env_nums != 1:
            raise _C.TrlError("BaseCollector drives a single env (torchrl.env.get_env); use VecCollector for %d envs"
                              % self.env.env_nums)
