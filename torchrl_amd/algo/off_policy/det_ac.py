"""DDPG and TD3 on the HIP path (reference: torchrl/algo/off_policy/ddpg.py:10-140, td3.py:10-190).

Both are deterministic actor-critic updates over MLPs, so they run on the same launch sequence as
TwinSACQ with an explicit chain rule instead of autograd:
  dense layers ............ trl_linear_{fwd,bwd_input,bwd_weight}_f32 (fp32 MFMA, k_gemm.hip); the policy's
                            tanh is the activation of its last layer's epilogue and gates its backward
  target smoothing (TD3) .. trl_noisy_action_f32 (explore noise of target_pf, then the clipped smoothing noise)
  TD target, MSE, dQ ...... trl_detac_losses_f32 (also -mean Q(s, pi(s)) and its gradient)
  dL/da through the critic  bwd_input of the Q net + trl_slice_add_f32
  clip + Adam ............. trl_clip_adam_f32 on one flat [pf | qf (| qf2)] buffer
  target update ........... trl_polyak_f32 on the flat buffers (policy AND critics, as the reference does)
The reference's order of operations is kept, including TD3's delayed policy step, which is taken when
`training_update_num % policy_update_delay != 0` (td3.py:124, as written) and evaluates Q1 AFTER its step.
TD3's two N(0,1) draws per update come from the CPU torch generator (reference parity) or from the device
Philox stream (`noise_mode="device"`).
"""
import copy
import os

import numpy as np
import torch
import torch.optim as optim

from ... import _C, dist, ops
from ...networks import flatten_into
from ._deferred import StatRing
from .off_rl_algo import OffRLAlgo


class DDPG(OffRLAlgo):
    def __init__(self, pf, qf, plr, qlr, optimizer_class=optim.Adam, **kwargs):
        super().__init__(**kwargs)
        self.pf, self.qf = pf, qf
        self.target_pf, self.target_qf = copy.deepcopy(pf), copy.deepcopy(qf)
        self.to(self.device)
        self.plr, self.qlr = plr, qlr
        self.optimizer_class = optimizer_class
        self.pf_optimizer = optimizer_class(self.pf.parameters(), lr=self.plr)
        self.qf_optimizer = optimizer_class(self.qf.parameters(), lr=self.qlr)
        self._engine = None

    @property
    def networks(self):
        return [self.pf, self.qf, self.target_pf, self.target_qf]

    @property
    def snapshot_networks(self):
        return [["pf", self.pf], ["qf", self.qf]]

    @property
    def target_networks(self):
        return [(self.pf, self.target_pf), (self.qf, self.target_qf)]

    def engine(self):
        if self._engine is None:
            self._engine = _FusedDetAC(self, [self.pf, self.qf], [self.target_pf, self.target_qf],
                                       [self.pf_optimizer, self.qf_optimizer])
        return self._engine

    def static_batch(self):
        return self.engine().static_batch(self.batch_size)

    def update(self, batch):
        self.training_update_num += 1
        return self.engine().update_ddpg(batch)

    def update_deferred(self, batch):
        """`update` without its read-back (OffRLAlgo.update_per_epoch resolves an epoch's updates with one D2H)."""
        self.training_update_num += 1
        return self.engine().enqueue_ddpg(batch)

    def resolve_updates(self, handles):
        return self.engine().resolve(handles)


class TD3(OffRLAlgo):
    def __init__(self, pf, qf1, qf2, plr, qlr, optimizer_class=optim.Adam, policy_update_delay=2,
                 norm_std_policy=0.2, noise_clip=0.5, noise_mode="host", **kwargs):
        super().__init__(**kwargs)
        self.pf, self.qf1, self.qf2 = pf, qf1, qf2
        self.target_pf, self.target_qf1, self.target_qf2 = copy.deepcopy(pf), copy.deepcopy(qf1), copy.deepcopy(qf2)
        self.to(self.device)
        self.plr, self.qlr = plr, qlr
        self.optimizer_class = optimizer_class
        self.pf_optimizer = optimizer_class(self.pf.parameters(), lr=self.plr)
        self.qf1_optimizer = optimizer_class(self.qf1.parameters(), lr=self.qlr)
        self.qf2_optimizer = optimizer_class(self.qf2.parameters(), lr=self.qlr)
        self.policy_update_delay = policy_update_delay
        self.norm_std_policy, self.noise_clip = norm_std_policy, noise_clip
        if noise_mode not in ("host", "device"):
            raise ValueError("noise_mode must be 'host' or 'device'")
        self.noise_mode = noise_mode
        self._engine = None

    @property
    def networks(self):
        return [self.pf, self.qf1, self.qf2, self.target_pf, self.target_qf1, self.target_qf2]

    @property
    def snapshot_networks(self):
        return [["pf", self.pf], ["qf1", self.qf1], ["qf2", self.qf2]]

    @property
    def target_networks(self):
        return [(self.pf, self.target_pf), (self.qf1, self.target_qf1), (self.qf2, self.target_qf2)]

    def engine(self):
        if self._engine is None:
            self._engine = _FusedDetAC(self, [self.pf, self.qf1, self.qf2],
                                       [self.target_pf, self.target_qf1, self.target_qf2],
                                       [self.pf_optimizer, self.qf1_optimizer, self.qf2_optimizer])
        return self._engine

    def static_batch(self):
        return self.engine().static_batch(self.batch_size)

    def update(self, batch):
        self.training_update_num += 1
        return self.engine().update_td3(batch)

    def update_deferred(self, batch):
        """`update` without its read-back (OffRLAlgo.update_per_epoch resolves an epoch's updates with one D2H)."""
        self.training_update_num += 1
        return self.engine().enqueue_td3(batch)

    def resolve_updates(self, handles):
        return self.engine().resolve(handles)


class _FusedDetAC:
    """Flat [pf | qf1 (| qf2)] parameter / gradient / Adam-state buffers and the two launch sequences."""

    def __init__(self, algo, nets, targets, optimizers):
        if algo.optimizer_class is not optim.Adam:
            raise _C.TrlError("the fused DDPG / TD3 step implements torch.optim.Adam only")
        self.algo = algo
        self.dev = next(nets[0].parameters()).device
        if self.dev.type != "cuda":
            raise _C.TrlError("networks live on %s: the HIP path needs a GPU (no CPU path exists)" % self.dev)
        self.act = ops.act_code(nets[0])
        if any(ops.act_code(n) != self.act for n in nets[1:]):
            raise _C.TrlError("policy and critics must use the same activation")
        self.pf_last = _C.ACT_TANH if getattr(nets[0], "tanh_action", False) else _C.ACT_NONE
        self.sigma_explore = float(getattr(nets[0], "norm_std_explore", 0.0))
        self.layers = [ops.linear_layers(n) for n in nets]
        self.tlayers = [ops.linear_layers(n) for n in targets]
        plists = [[t for wb in ls for t in wb] for ls in self.layers]
        self.sizes = [sum(p.numel() for p in pl) for pl in plists]
        self.flat = flatten_into([p for pl in plists for p in pl])
        self.tflat = flatten_into([t for ls in self.tlayers for wb in ls for t in wb])
        self.grads = torch.zeros_like(self.flat)
        self.m, self.v = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        self.gviews, off = [], 0
        for ls in self.layers:
            views = []
            for w, b in ls:
                gw = self.grads[off:off + w.numel()].view(w.shape); off += w.numel()
                gb = self.grads[off:off + b.numel()].view(b.shape); off += b.numel()
                views.append((gw, gb))
            self.gviews.append(views)
        self.offsets = np.concatenate([[0], np.cumsum(self.sizes)]).astype(int)
        off = 0
        for opt, pl in zip(optimizers, plists):
            for p in pl:
                n = p.numel()
                opt.state[p] = {"step": torch.tensor(0.0), "exp_avg": self.m[off:off + n].view(p.shape),
                                "exp_avg_sq": self.v[off:off + n].view(p.shape)}
                off += n
        self.optimizers = optimizers
        self.steps = [0] * len(nets)                                       # per-network Adam step counts (TD3's policy lags)
        self._raw = torch.zeros(96 + 16, dtype=torch.uint8, device=self.dev)   # every logged statistic: one D2H per update
        self.sums = self._raw[0:32].view(torch.float64)
        self.sums_p = self._raw[32:64].view(torch.float64)
        self.mom = self._raw[64:96].view(torch.float64)
        self.norms = self._raw[96:96 + 4 * len(nets)].view(torch.float32)
        # per-network Adam step state on the device {steps, beta1^steps, beta2^steps, 0}: a captured graph replays unchanged
        self.step_state = torch.tensor([[0.0, 1.0, 1.0, 0.0]] * len(nets), dtype=torch.float64, device=self.dev)
        self._static, self._graphs, self._seen = {}, {}, set()
        self.workspace = None
        self.D = int(self.layers[0][0][0].shape[1])
        self.A = int(self.layers[0][-1][0].shape[0])
        self.noise_ctr, self.noise_seed = 0, 0x7D3

    # ---- helpers ----
    def _ws(self, B):
        need = 2 * max(_C.lib().trl_linear_bwd_weight_workspace(B, int(w.shape[1]), int(w.shape[0]))
                       for ls in self.layers for w, _ in ls)               # x2: the twin critics' grouped weight gradient
        if self.workspace is None or self.workspace.numel() < need:
            self.workspace = torch.empty(need, device=self.dev)
        return self.workspace

    def static_batch(self, B):
        """Persistent input tensors of a B-row update (`random_batch(..., out=...)` gathers straight into them)."""
        st = self._static.get(B)
        if st is None:
            f = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=self.dev)
            st = {"obs": f(B, self.D), "next_obs": f(B, self.D), "acts": f(B, self.A), "rewards": f(B, 1),
                  "terminals": f(B, 1), "eps_explore": f(B, self.A), "eps_smooth": f(B, self.A)}
            self._static[B] = st
        return st

    def _load(self, batch, noise_keys):
        B = int(batch['obs'].shape[0])
        st = self.static_batch(B)
        for k in ("obs", "next_obs", "acts", "rewards", "terminals"):
            src = batch[k]
            if src is st[k]:
                continue
            src = src if isinstance(src, torch.Tensor) else torch.as_tensor(np.asarray(src))
            st[k].copy_(src.to(dtype=torch.float32).reshape(st[k].shape), non_blocking=True)
        buf = getattr(self.algo, "replay_buffer", None)
        n_env = int(buf.env_nums) if buf is not None and B % int(buf.env_nums) == 0 else B
        for k in noise_keys:                                               # this rank's block of the draw for all envs
            if getattr(self.algo, "noise_mode", "host") == "host":         # CPU generator draw (reference stream)
                make = lambda m, f: torch.randn(m, f)
            else:
                self.noise_ctr += 1
                make = lambda m, f: _C.philox_normal(torch.empty(m, f, device=self.dev), self.noise_seed, self.noise_ctr)
            st[k].copy_(dist.shard_rows_of_global(make, B // n_env, n_env, self.A, self.dev), non_blocking=True)
        return st, B

    def _lrs(self):
        return tuple(float(o.param_groups[0]['lr']) for o in self.optimizers)

    def _run(self, key, seq):
        """Eager on the first visit of a configuration, captured into a HIP graph on the second, replayed afterwards
        (TRL_NO_GRAPH=1 keeps everything eager)."""
        algo = self.algo
        key = key + (self._lrs(), algo.grad_clip, algo.tau, algo.discount)
        if os.environ.get("TRL_NO_GRAPH") == "1" or dist.collectives_active() or \
                (key not in self._graphs and len(self._graphs) >= 8):
            seq()                                                            # (a learning-rate schedule would mint a key per value)
        elif key in self._graphs:
            self._graphs[key].replay()
        elif key not in self._seen:
            self._seen.add(key)
            seq()
        else:
            graph, _ = _C.capture_graph(seq)
            self._graphs[key] = graph
            graph.replay()

    def _adam(self, which):
        """clip_grad_norm_ + Adam for the networks in `which` (indices into the flat buffer), one launch each;
        every network keeps its own step count, on the device."""
        algo = self.algo
        for k in which:
            dist.all_reduce_sum_(self.grads[int(self.offsets[k]):int(self.offsets[k + 1])])   # C1: local means -> SUM / world
            a = _C.AdamArgs()
            o = int(self.offsets[k]) * 4
            a.params, a.grads = self.flat.data_ptr() + o, self.grads.data_ptr() + o
            a.exp_avg, a.exp_avg_sq = self.m.data_ptr() + o, self.v.data_ptr() + o
            a.n_groups = 1
            a.group_sizes[0] = self.sizes[k]
            a.group_lr[0] = self.optimizers[k].param_groups[0]['lr']
            a.max_norm = float(algo.grad_clip) if algo.grad_clip else 0.0
            a.beta1, a.beta2, a.eps, a.grad_scale = 0.9, 0.999, 1e-8, 1.0 / dist.world_size()
            a.step_count, a.norms_out = 0, self.norms.data_ptr() + 4 * k
            a.step_state = self.step_state.data_ptr() + 32 * k
            _C.clip_adam(a, self.dev)

    def _hard_update_due(self):
        algo = self.algo
        return (not algo.use_soft_update) and algo.training_update_num % algo.target_hard_update_period == 0

    def _policy_grad(self, obs, q_layers):
        """Q(s, pi(s)) with the tapes of both networks."""
        new_a, tape_pf = ops.mlp_forward(self.layers[0], obs, self.act, last_act=self.pf_last)
        qn, tape_qn = ops.mlp_forward(q_layers, _C.concat2(obs, new_a), self.act)
        return new_a, tape_pf, qn, tape_qn

    @staticmethod
    def _flat(st):
        return st["obs"], st["acts"], st["next_obs"], st["rewards"].view(-1), st["terminals"].view(-1)

    # ---- DDPG (ddpg.py:42-110) ----
    def _seq_ddpg(self, st, soft):
        algo, D, A = self.algo, self.D, self.A
        obs, acts, nobs, rew, term = self._flat(st)
        ws = self._ws(int(obs.shape[0]))
        pf_l, qf_l = self.layers
        # policy on obs and target policy on next_obs, then the three critic passes, each as one grouped launch per layer
        (new_a, ta), (tape_pf, _) = ops.mlp_forward_group([pf_l, self.tlayers[0]], [obs, nobs], self.act,
                                                          last_act=self.pf_last)
        (qn, tq, qp), (tape_qn, _, tape_q) = ops.mlp_forward_group(
            [qf_l, self.tlayers[1], qf_l], [_C.concat2(obs, new_a), _C.concat2(nobs, ta), _C.concat2(obs, acts)], self.act)
        dq, _, dqn = _C.detac_losses(qp, None, tq, None, rew, term, qn, algo.discount, self.sums)
        dx = ops.mlp_backward(tape_qn, dqn, grads=None, need_input=True)
        ops.mlp_backward(tape_pf, _C.slice_add(dx, None, D, A), grads=self.gviews[0], workspace=ws)
        ops.mlp_backward(tape_q, dq, grads=self.gviews[1], workspace=ws)
        self._adam((0, 1))
        if soft:
            _C.polyak(self.tflat, self.flat, algo.tau)
        dist.all_reduce_sum_(self.sums)
        _C.moments(dist.all_gather_cat(new_a), self.mom, ld=1)

    # ---- launch now, read back later: the statistics of an update go (stream-ordered) into a slot of a device ring ----
    def _park(self, *extra):
        if getattr(self, "_ring", None) is None:
            self._ring = StatRing(max(64, int(getattr(self.algo, "opt_times", 1))), self._raw.numel(), torch.uint8, self.dev)
            self._parked = 0
        self._ring.make_room(1)
        (ref, row), = self._ring.handles(self._parked, 1)
        self._parked += 1
        self._ring.t[row].copy_(self._raw, non_blocking=True)
        return (ref, row) + extra

    def resolve(self, handles):
        """Info dicts of enqueued updates, in order, after one D2H per ring (the only host sync)."""
        if not handles:
            return []
        rows = torch.from_numpy(self._ring.read([(h[0], h[1]) for h in handles]))
        return [h[2](raw, *h[3:]) for raw, h in zip(rows, handles)]

    def update_ddpg(self, batch):
        return self.resolve([self.enqueue_ddpg(batch)])[0]

    def update_td3(self, batch):
        return self.resolve([self.enqueue_td3(batch)])[0]

    def enqueue_ddpg(self, batch):
        algo = self.algo
        st, B = self._load(batch, ())
        soft = bool(algo.use_soft_update)
        self._run(("ddpg", B, soft), lambda: self._seq_ddpg(st, soft))
        self.steps = [n + 1 for n in self.steps]
        if self._hard_update_due():
            _C.polyak(self.tflat, self.flat, 1.0)
        return self._park(self._info_ddpg, B * dist.world_size())           # the sums cover every rank's samples

    def _info_ddpg(self, raw, B):
        algo = self.algo
        sums, m = raw[0:32].view(torch.float64).numpy(), raw[64:96].view(torch.float64).numpy()
        norms = raw[96:].view(torch.float32).numpy()
        info = {'Reward_Mean': sums[3] / B, 'Training/policy_loss': sums[2] / B, 'Training/qf_loss': sums[0] / B}
        if algo.grad_clip is not None:
            info['Training/pf_grad_norm'], info['Training/qf_grad_norm'] = float(norms[0]), float(norms[1])
        info['new_actions/mean'], info['new_actions/std'], info['new_actions/max'], info['new_actions/min'] = \
            float(m[0]), float(m[1]), float(m[2]), float(m[3])
        return info

    # ---- TD3 (td3.py:57-154) ----
    def _seq_td3(self, st, delayed, soft):
        algo, D, A = self.algo, self.D, self.A
        obs, acts, nobs, rew, term = self._flat(st)
        ws = self._ws(int(obs.shape[0]))
        pf_l, q1_l, q2_l = self.layers
        ta, _ = ops.mlp_forward(self.tlayers[0], nobs, self.act, last_act=self.pf_last)
        if self.sigma_explore:
            ta = _C.noisy_action(ta, st["eps_explore"], self.sigma_explore)
        ta = _C.noisy_action(ta, st["eps_smooth"], algo.norm_std_policy, algo.noise_clip, -1.0, 1.0)
        x_next, x_sa = _C.concat2(nobs, ta), _C.concat2(obs, acts)
        # target Q1 / Q2 (s', a') and Q1 / Q2 (s, a): four same-shaped critics, one grouped launch per layer
        (tq1, tq2, q1p, q2p), (_, _, tape_q1, tape_q2) = ops.mlp_forward_group(
            [self.tlayers[1], self.tlayers[2], q1_l, q2_l], [x_next, x_next, x_sa, x_sa], self.act)
        dq1, dq2, _ = _C.detac_losses(q1p, q2p, tq1, tq2, rew, term, None, algo.discount, self.sums)
        ops.mlp_backward_group([tape_q1, tape_q2], [dq1, dq2], grads_list=[self.gviews[1], self.gviews[2]], workspace=ws)
        self._adam((1, 2))
        if delayed:                                                          # policy step on the UPDATED Q1
            new_a, tape_pf, qn, tape_qn = self._policy_grad(obs, q1_l)
            _, _, dqn = _C.detac_losses(q1p, None, tq1, None, rew, term, qn, algo.discount, self.sums_p)
            dx = ops.mlp_backward(tape_qn, dqn, grads=None, need_input=True)
            ops.mlp_backward(tape_pf, _C.slice_add(dx, None, D, A), grads=self.gviews[0], workspace=ws)
            self._adam((0,))
            if soft:
                _C.polyak(self.tflat, self.flat, algo.tau)
            dist.all_reduce_sum_(self.sums_p)
            _C.moments(dist.all_gather_cat(new_a), self.mom, ld=1)
        dist.all_reduce_sum_(self.sums)

    def enqueue_td3(self, batch):
        algo = self.algo
        # target_pf.explore draws only for policies with exploration noise; then the smoothing draw
        st, B = self._load(batch, (("eps_explore",) if self.sigma_explore else ()) + ("eps_smooth",))
        delayed = bool(algo.training_update_num % algo.policy_update_delay)
        soft = bool(algo.use_soft_update)
        self._run(("td3", B, delayed, soft), lambda: self._seq_td3(st, delayed, soft))
        self.steps = [self.steps[0] + int(delayed), self.steps[1] + 1, self.steps[2] + 1]
        if delayed and self._hard_update_due():
            _C.polyak(self.tflat, self.flat, 1.0)
        return self._park(self._info_td3, B * dist.world_size(), delayed)   # the sums cover every rank's samples

    def _info_td3(self, raw, B, delayed):
        algo = self.algo
        sums, sums_p = raw[0:32].view(torch.float64).numpy(), raw[32:64].view(torch.float64).numpy()
        m, norms = raw[64:96].view(torch.float64).numpy(), raw[96:].view(torch.float32).numpy()
        info = {'Reward_Mean': sums[3] / B, 'Training/qf1_loss': sums[0] / B, 'Training/qf2_loss': sums[1] / B}
        if algo.grad_clip is not None:
            info['Training/qf1_grad_norm'], info['Training/qf2_grad_norm'] = float(norms[1]), float(norms[2])
        if delayed:
            info['Training/policy_loss'] = float(sums_p[2]) / B
            if algo.grad_clip is not None:
                info['Training/pf_grad_norm'] = float(norms[0])
            info['new_actions/mean'], info['new_actions/std'], info['new_actions/max'], info['new_actions/min'] = \
                float(m[0]), float(m[1]), float(m[2]), float(m[3])
        return info
