"""Twin-Q SAC (no V net) on the HIP path (reference: torchrl/algo/off_policy/twin_sac_q.py:10-244).

`update(batch)` keeps the reference's order of operations -- sample new actions, alpha step (alpha
is re-read AFTER it), no-grad target branch, twin MSE, policy loss through min(Q1, Q2) of the new
actions, optimiser steps pf -> qf1 -> qf2, Polyak -- but runs as a fixed launch sequence with an
explicit chain rule instead of autograd:
  dense layers ............ trl_linear_{fwd,bwd_input,bwd_weight}[_group]_f32   (fp32 MFMA, k_gemm.hip); the six
                            critic passes, the two policy passes and the twin backward passes are grouped launches
  rsample / its backward .. trl_tanh_gauss_rsample_{fwd,bwd}_f32
  alpha loss + its Adam ... trl_sac_alpha_step_f32 (log_alpha, its moments and alpha stay on device)
  TD target, losses, dQ ... trl_sac_losses_f32
  dL/da through both Qs ... bwd_input of the two Q nets + trl_slice_add_f32
  clip + Adam x3 .......... trl_clip_adam_f32 on one flat [pf | qf1 | qf2] buffer
  target update ........... trl_polyak_f32 on flat [qf1 | qf2] -> [tqf1 | tqf2]
policy_loss.backward() in the reference also deposits gradients in qf1/qf2 that the following
zero_grad() discards (its Q20); only d/d(action) is propagated here.
The two N(0,1) draws per update come from the CPU torch generator (reference parity) or from the
device Philox stream (`noise_mode="device"`).
"""
import copy
import os

import numpy as np
import torch
import torch.optim as optim

from ... import _C, dist, ops
from ...networks import flatten_into
from ._deferred import IndexSlab, StatRing
from .off_rl_algo import OffRLAlgo


class TwinSACQ(OffRLAlgo):
    def __init__(self, pf, qf1, qf2, plr, qlr, optimizer_class=optim.Adam, policy_std_reg_weight=1e-3,
                 policy_mean_reg_weight=1e-3, reparameterization=True, automatic_entropy_tuning=True,
                 target_entropy=None, noise_mode="host", **kwargs):
        super().__init__(**kwargs)
        self.pf, self.qf1, self.qf2 = pf, qf1, qf2
        self.target_qf1 = copy.deepcopy(qf1)
        self.target_qf2 = copy.deepcopy(qf2)
        self.to(self.device)
        self.plr, self.qlr = plr, qlr
        self.optimizer_class = optimizer_class
        self.qf1_optimizer = optimizer_class(self.qf1.parameters(), lr=self.qlr)
        self.qf2_optimizer = optimizer_class(self.qf2.parameters(), lr=self.qlr)
        self.pf_optimizer = optimizer_class(self.pf.parameters(), lr=self.plr)
        self.automatic_entropy_tuning = automatic_entropy_tuning
        if target_entropy:
            self.target_entropy = target_entropy
        else:
            self.target_entropy = -float(np.prod(self.env.action_space.shape))
        self.policy_std_reg_weight = policy_std_reg_weight
        self.policy_mean_reg_weight = policy_mean_reg_weight
        if not reparameterization:
            raise NotImplementedError
        self.reparameterization = reparameterization
        if noise_mode not in ("host", "device"):
            raise ValueError("noise_mode must be 'host' or 'device'")
        self.noise_mode = noise_mode
        self._engine = None

    @property
    def networks(self):
        return [self.pf, self.qf1, self.qf2, self.target_qf1, self.target_qf2]

    @property
    def snapshot_networks(self):
        return [["pf", self.pf], ["qf1", self.qf1], ["qf2", self.qf2]]

    @property
    def target_networks(self):
        return [(self.qf1, self.target_qf1), (self.qf2, self.target_qf2)]

    @property
    def log_alpha(self):
        return self.engine().alpha_state[0:1]

    def engine(self):
        if self._engine is None:
            if self.optimizer_class is not optim.Adam:
                raise _C.TrlError("the fused SAC step implements torch.optim.Adam only")
            self._engine = _FusedSAC(self)
        return self._engine

    def static_batch(self):
        return self.engine().static_batch(self.batch_size)

    def update(self, batch):
        self.training_update_num += 1
        return self.engine().update(batch)

    def update_deferred(self, batch):
        """`update` without its read-back: OffRLAlgo.update_per_epoch enqueues all `opt_times` updates of an epoch and
        resolves their info dicts with one D2H (`resolve_updates`)."""
        self.training_update_num += 1
        return self.engine().enqueue(batch)

    def update_epoch_deferred(self, count):
        """All `count` {uniform sample -> update} pairs of an epoch as one replayed graph (None when the engine's
        conditions for it do not hold: the caller then samples and enqueues one by one)."""
        handles = self.engine().enqueue_epoch(count)
        if handles is not None:
            self.training_update_num += count
        return handles

    def resolve_updates(self, handles):
        return self.engine().resolve(handles)


class _FusedSAC:
    def __init__(self, algo):
        self.algo = algo
        nets = (algo.pf, algo.qf1, algo.qf2)
        self.dev = next(algo.pf.parameters()).device
        if self.dev.type != "cuda":
            raise _C.TrlError("TwinSACQ networks live on %s: the HIP path needs a GPU (no CPU path exists)" % self.dev)
        self.act = ops.act_code(algo.pf)
        if any(ops.act_code(n) != self.act for n in nets[1:]):
            raise _C.TrlError("pf / qf1 / qf2 must use the same activation")
        self.layers = [ops.linear_layers(n) for n in nets]
        plists = [[t for wb in ls for t in wb] for ls in self.layers]
        self.sizes = [sum(p.numel() for p in pl) for pl in plists]
        self.flat = flatten_into(plists[0] + plists[1] + plists[2])       # [pf | qf1 | qf2]; parameters become views
        self.tlayers = [ops.linear_layers(n) for n in (algo.target_qf1, algo.target_qf2)]
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
        off = 0
        for opt, pl in zip((algo.pf_optimizer, algo.qf1_optimizer, algo.qf2_optimizer), plists):
            for p in pl:
                n = p.numel()
                opt.state[p] = {"step": torch.tensor(0.0), "exp_avg": self.m[off:off + n].view(p.shape),
                                "exp_avg_sq": self.v[off:off + n].view(p.shape)}
                off += n
        self.step_count = 0
        self.alpha_state = torch.zeros(4, device=self.dev)               # log_alpha, exp_avg, exp_avg_sq, step
        self._raw = torch.zeros(160, dtype=torch.uint8, device=self.dev)  # every logged statistic, one D2H per update
        self.sums = self._raw[:32].view(torch.float64)
        self.mom = self._raw[32:128].view(torch.float64).view(3, 4)
        self.alpha_out = self._raw[128:136].view(torch.float32)         # alpha, alpha_loss
        self.alpha_out.fill_(1.0)
        self.norms = self._raw[136:148].view(torch.float32)
        self.workspace = None
        self.D = int(self.layers[0][0][0].shape[1])
        self.A = int(self.layers[0][-1][0].shape[0]) // 2
        self.noise_ctr = 0
        self.noise_seed = 0x5AC
        self.step_state = torch.tensor([0.0, 1.0, 1.0, 0.0], dtype=torch.float64, device=self.dev)
        self._static, self._graphs, self._seen = {}, {}, set()
        # the statistics block of update u is filed into ring row u % slots by the update's last launch
        self._ring = StatRing(max(64, int(getattr(algo, "opt_times", 1))), self._raw.numel(), torch.uint8, self.dev)
        self._slab = IndexSlab(self.dev)
        self._mom_part = None
        self._pg_fused = True                                           # policy gradient: sac_policy_grad (False: layer GEMM + sampler launch)
        self._fused_tail, self._tail_ws = True, None                    # fold + clip + Adam + Polyak as one launch

    def _ws(self, B):
        lib = _C.lib()
        need = sum(max(lib.trl_linear_bwd_weight_workspace(B, int(w.shape[1]), int(w.shape[0])),
                       lib.trl_linear_bwd_weight_multi_splits(B, int(w.shape[1]), int(w.shape[0])) * (w.numel() + int(w.shape[0])))
                   for ls in self.layers for w, _ in ls)                  # every layer's partials live until the one fold
        if self.workspace is None or self.workspace.numel() < need:
            self.workspace = torch.empty(need, device=self.dev)
        return self.workspace

    def static_batch(self, B):
        """Persistent input tensors of a B-row update (the replay gather can write straight into them:
        `random_batch(..., out=engine.static_batch(B))`); a captured graph reads these addresses."""
        st = self._static.get(B)
        if st is None:
            f = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=self.dev)
            st = {"obs": f(B, self.D), "next_obs": f(B, self.D), "acts": f(B, self.A), "rewards": f(B, 1),
                  "terminals": f(B, 1), "eps1": f(B, self.A), "eps2": f(B, self.A)}
            self._static[B] = st
        return st

    def _sequence(self, st, soft):
        """The fixed launch sequence of one update on the static inputs (eager, or under graph capture)."""
        algo, dev, A, D = self.algo, self.dev, self.A, self.D
        obs, acts, nobs = st["obs"], st["acts"], st["next_obs"]
        rew, term = st["rewards"].view(-1), st["terminals"].view(-1)
        eps1, eps2 = st["eps1"], st["eps2"]
        B = int(obs.shape[0])
        ws = self._ws(B)
        pf_l, q1_l, q2_l = self.layers
        tanh_action = bool(algo.pf.tanh_action)
        # ---- policy on obs and next_obs (one grouped launch per layer), both samples ----
        (head, head2), (tape_pf, _) = ops.mlp_forward_group([pf_l, pf_l], [obs, nobs], self.act, keep=[True, False])
        # both samples (distribution.py:67-70 order) and the three critic inputs [obs | acts], [next_obs | next_a],
        # [obs | new_a]: one launch
        # (device noise on one rank: the two draws are made inside that launch from the device-resident update count)
        # One rank with soft target updates: the temperature step, the logged moments and the filing of the statistics block
        # ride on launches the update has anyway (sampling -> partial moments, loss launch -> temperature step + fold of
        # the partials, Polyak launch -> filing); otherwise they are the separate launches sac_alpha / moments_multi.
        ride = self._stats_ride_along(soft)
        if ride and (self._mom_part is None or self._mom_part.shape[0] != (B + 63) // 64):
            self._mom_part = torch.zeros((B + 63) // 64, 12, dtype=torch.float64, device=dev)
        new_a, logp, next_a, next_logp, x_sa, x_next, x_new = _C.sac_samples(
            head, head2, eps1, eps2, obs, acts, nobs, tanh_action,
            philox=(self.step_state, self.noise_seed) if self._inline_noise() else None,
            mom_part=self._mom_part if ride else None)
        # ---- temperature ----
        if algo.automatic_entropy_tuning and not ride:                   # mean over the GLOBAL batch (all ranks' samples)
            _C.sac_alpha_step(dist.all_gather_cat(logp), algo.target_entropy, algo.plr, self.alpha_state, self.alpha_out)
        # ---- all six critic passes as one group: Q1/Q2(s, a), target Q1/Q2(s', a'), Q1/Q2(s, new a) ----
        (q1p, q2p, tq1, tq2, q1n, q2n), (tape_q1, tape_q2, _, _, tape_q1n, tape_q2n) = ops.mlp_forward_group(
            [q1_l, q2_l, self.tlayers[0], self.tlayers[1], q1_l, q2_l], [x_sa, x_sa, x_next, x_next, x_new, x_new], self.act,
            keep=[True, True, False, False, True, True])                 # the target passes leave no tape
        alpha = self.alpha_out[0:1]                                      # re-read AFTER the alpha step, as the reference does
        dq1, dq2, dq1n, dq2n = _C.sac_losses(
            q1p, q2p, tq1, tq2, next_logp, rew, term, q1n, q2n, logp, alpha, algo.discount, self.sums,
            alpha_step=(algo.target_entropy, algo.plr, self.alpha_state, self.alpha_out)
            if ride and algo.automatic_entropy_tuning else None,
            fold=(self._mom_part, A, self.mom) if ride else None)
        # weight gradients of all nine layers: ONE launch of split GEMMs + ONE fold, after the input-gradient chains
        plan = _C.FoldPlan(ws, defer_gemm=True)
        # ---- the four critic backward passes as one group: through Q1 / Q2 on [obs | new_a] down to the action (policy
        # gradient, no weight gradients: twin_sac_q.py:152-155 discards them, its Q20) and the Q-loss pass on [obs | acts]
        # (weight gradients, no input gradient) ----
        # The first critic layer's input gradient is only needed in its A action columns, summed over the twins and pushed
        # through the sampler's backward: one streaming launch (sac_policy_grad) instead of a (B x 256) . (256 x (D + A)) GEMM
        # per critic and the sampler launch behind it.
        first, pol_dz = [], None
        dx1, dx2, _, _ = ops.mlp_backward_group([tape_q1n, tape_q2n, tape_q1, tape_q2], [dq1n, dq2n, dq1, dq2],
                                                grads_list=[None, None, self.gviews[1], self.gviews[2]],
                                                need_input=[True, True, False, False], plan=plan,
                                                input_sink=(lambda *a: first.append(a)) if self._pg_fused else None)
        if first:
            dys, ys, gate_act, w0 = first[0]
            if _C.sac_policy_grad_ok(dys, w0, A):
                # ... and the policy's own head backward rides along: dZ2 = (d_head W3) * act'(H2) leaves the same launch
                # already gated (a 2A-deep input-gradient GEMM launch less; the layer below reads one operand, not two)
                hl = (pf_l[-1][0], tape_pf.outs[-2], self.act) if len(pf_l) >= 2 and len(tape_pf.outs) >= 2 else None
                d_head = _C.sac_policy_grad(head, eps1, new_a, dys, ys, gate_act, w0, D, alpha, 1.0 / B,
                                            algo.policy_std_reg_weight, algo.policy_mean_reg_weight, tanh_action,
                                            head_layer=hl)
                if hl is not None:
                    d_head, pol_dz = d_head
            else:                                                         # shapes outside the streaming kernel: the layer GEMM
                dx1, dx2 = _C.linear_bwd_input_group(dys, ys, gate_act, w0)
                first = []
        if not first:
            d_head = _C.rsample_bwd_cols(head, eps1, new_a, dx1, dx2, D, alpha, 1.0 / B, algo.policy_std_reg_weight,
                                         algo.policy_mean_reg_weight, tanh_action)    # d_act = (dx1 + dx2)[:, D:]
        ops.mlp_backward(tape_pf, d_head, grads=self.gviews[0], plan=plan, head_dx=pol_dz)
        # one process, soft target updates: the folds, the clip, the Adam steps, the Polyak step and the filing of the
        # statistics are ONE launch (FoldPlan.run_fused); otherwise fold here, (all-reduce,) clip + Adam (+ Polyak) below
        fused_tail = soft and self._fused_tail and dist.world_size() == 1 and plan.tiles(self.grads)
        if not fused_tail:
            plan.run()
        # ---- optimiser steps (pf, qf1, qf2) and target update ----
        a = _C.AdamArgs()
        a.params, a.grads, a.exp_avg, a.exp_avg_sq = (self.flat.data_ptr(), self.grads.data_ptr(),
                                                      self.m.data_ptr(), self.v.data_ptr())
        a.n_groups = 3
        for k in range(3):
            a.group_sizes[k] = self.sizes[k]
        for k, lr in enumerate(self._lrs()):
            a.group_lr[k] = lr
        a.max_norm = float(algo.grad_clip) if algo.grad_clip else 0.0
        a.beta1, a.beta2, a.eps = 0.9, 0.999, 1e-8
        dist.all_reduce_sum_(self.grads)                                 # C1: every loss is a local mean -> SUM / world
        a.grad_scale = 1.0 / dist.world_size()
        a.step_count, a.norms_out = 0, self.norms.data_ptr()
        a.step_state = self.step_state.data_ptr()                        # the step count lives on the device
        if fused_tail:
            if self._tail_ws is None:
                self._tail_ws = _C.fold_clip_adam_polyak_workspace(dev)
            plan.run_fused(a, self.grads, self.tflat, self.sizes[0], algo.tau, self._tail_ws,
                           file=(self._raw, self._ring.t) if ride else None)
        elif soft:                                                       # Adam, then Polyak (which also advances the step state)
            _C.clip_adam_polyak(a, self.tflat, self.flat[self.sizes[0]:], algo.tau, dev,
                                file=(self._raw, self._ring.t) if ride else None)
        else:
            _C.clip_adam(a, dev)
        if ride:
            return
        # ---- logging statistics (over the global batch) ----
        dist.all_reduce_sum_(self.sums)
        head_g, logp_g = dist.all_gather_cat(head), dist.all_gather_cat(logp)
        inf = float("inf")
        _C.moments_multi([(head_g, self.mom[0], 2 * A, A, A, -20.0, 2.0),              # clamped log_std
                          (logp_g, self.mom[1], 1, 0, 1, -inf, inf),
                          (head_g, self.mom[2], 2 * A, 0, A, -inf, inf)],
                         ring=(self._raw, self._ring.t, self.step_state))                # ... and files the update's numbers

    def _stats_ride_along(self, soft):
        """One rank with soft target updates: temperature step, logged moments and the filing of the statistics block ride
        on launches the update has anyway; several ranks pool log pi / the moments first, so they keep their own launches."""
        return dist.world_size() == 1 and soft

    def _inline_noise(self):
        """Device Philox noise on a single rank whose every update so far drew its noise on the device: draw u's counters
        (2u + 1, 2u + 2) follow from the optimiser step count, so the sampling launch makes them itself."""
        return self.algo.noise_mode == "device" and dist.world_size() == 1 and self.noise_ctr == 2 * self.step_count

    def _lrs(self):
        algo = self.algo
        return tuple(float(o.param_groups[0]['lr']) for o in (algo.pf_optimizer, algo.qf1_optimizer, algo.qf2_optimizer))

    def _run(self, st, soft):
        """Eager on the first visit of a configuration, captured into a HIP graph on the second, replayed
        afterwards: ~90 dependent launches per update otherwise pay the eager launch latency each
        (TRL_NO_GRAPH=1 keeps everything eager)."""
        key = (int(st["obs"].shape[0]), soft, self._lrs(), self.algo.grad_clip, self.algo.tau, self.algo.discount,
               bool(self.algo.automatic_entropy_tuning), self._inline_noise())
        if os.environ.get("TRL_NO_GRAPH") == "1" or dist.collectives_active() or \
                (key not in self._graphs and len(self._graphs) >= 8):     # collectives between ranks: eager launches
            self._sequence(st, soft)                                     # (a learning-rate schedule would mint a key per value)
        elif key in self._graphs:
            self._graphs[key].replay()
        elif key not in self._seen:
            self._seen.add(key)
            self._sequence(st, soft)
        else:
            graph, _ = _C.capture_graph(lambda: self._sequence(st, soft))
            self._graphs[key] = graph
            graph.replay()

    def enqueue(self, batch):
        """Launch one update without waiting for it; returns the handle `resolve` turns into the info dict.  The logged
        statistics of the update are copied (stream-ordered) into a slot of a device ring, so that an epoch's
        `opt_times` updates need ONE read-back instead of one host sync each."""
        algo, dev, A, D = self.algo, self.dev, self.A, self.D
        B = int(batch['obs'].shape[0])
        st = self.static_batch(B)
        for k in ("obs", "next_obs", "acts", "rewards", "terminals"):
            src = batch[k]
            if src is st[k]:
                continue
            src = src if isinstance(src, torch.Tensor) else torch.as_tensor(np.asarray(src))
            st[k].copy_(src.to(dtype=torch.float32).reshape(st[k].shape), non_blocking=True)
        n_env = int(algo.replay_buffer.env_nums) if getattr(algo, "replay_buffer", None) is not None else B
        rows = B // n_env if B % n_env == 0 else 1                       # batch = sampled time rows x this rank's envs
        inline = self._inline_noise()
        for k in ("eps1", "eps2"):                                       # distribution.py:67-70: two draws, in this order
            if inline:
                continue
            if algo.noise_mode == "host":
                make = lambda m, f: torch.randn(m, f)                    # the CPU generator (reference stream)
            else:
                self.noise_ctr += 1
                if dist.world_size() == 1:                               # straight into the graph's input
                    _C.philox_normal(st[k], self.noise_seed, self.noise_ctr)
                    continue
                make = lambda m, f: _C.philox_normal(torch.empty(m, f, device=dev), self.noise_seed, self.noise_ctr)
            st[k].copy_(dist.shard_rows_of_global(make, rows, B // rows, A, dev), non_blocking=True)
        self._ring.make_room(1)
        self._run(st, bool(algo.use_soft_update))
        if inline:
            self.noise_ctr += 2                                          # the two draws the sampling launch made
        self.step_count += 1
        if not algo.use_soft_update and algo.training_update_num % algo.target_hard_update_period == 0:
            _C.polyak(self.tflat, self.flat[self.sizes[0]:], 1.0)
        return self._handles(1, B)[0]

    def _handles(self, count, B):
        """Handles of the `count` updates just launched (`step_count` mirrors the device-resident count the filing launch
        reads)."""
        return [(ref, row, B) for ref, row in self._ring.handles(self.step_count - count, count)]

    def enqueue_epoch(self, count):
        """`count` x {uniform replay sample -> update} as ONE graph launch: the index sets are drawn on the host in the
        order `count` random_batch calls would draw them and uploaded once; every update of the graph gathers the set
        the device-resident update count selects (trl_gather_rows_multi with its update counter), draws its own noise from that count and files
        its statistics into its ring slot.  Returns the handles, or None when this path does not apply (host noise,
        several ranks, hard target updates, a replay buffer that does not store the sampled keys plainly, TRL_NO_GRAPH=1)."""
        algo = self.algo
        buf, B = getattr(algo, "replay_buffer", None), int(algo.batch_size)
        if buf is None or count < 1 or count > self._ring.slots or not algo.use_soft_update or \
                not self._inline_noise() or dist.collectives_active() or os.environ.get("TRL_NO_GRAPH") == "1" or \
                not hasattr(buf, "gather_sources"):
            return None
        keys = ("obs", "next_obs", "acts", "rewards", "terminals")
        srcs = buf.gather_sources(keys)
        st = self.static_batch(B)
        if srcs is None or any(s.dtype != torch.float32 or s[0].numel() * buf._rows_per_batch(B) != st[k].numel()
                               for s, k in zip(srcs, keys)):
            return None
        nrows = buf._rows_per_batch(B)
        key = ("epoch", count, B, nrows, tuple(s.data_ptr() for s in srcs), self._lrs(), algo.grad_clip, algo.tau,
               algo.discount, bool(algo.automatic_entropy_tuning))
        if key not in self._graphs and len(self._graphs) >= 8:
            return None
        self._ring.make_room(count)
        slab = self._slab.upload(self.step_count, buf.draw_indices(B, count))
        dsts = [st[k] for k in keys]

        def launches(n):
            for _ in range(n):
                _C.gather_rows_multi(srcs, slab, dsts, slab_counter=self.step_state, n_rows=nrows)
                self._sequence(st, True)
        if key in self._graphs:
            self._graphs[key].replay()
        elif key not in self._seen:                                      # first visit eager, captured on the second
            self._seen.add(key)
            launches(count)
        else:
            graph, _ = _C.capture_graph(lambda: launches(count))
            self._graphs[key] = graph
            graph.replay()
        self.noise_ctr += 2 * count
        self.step_count += count
        return self._handles(count, B)

    def resolve(self, handles):
        """Info dicts of enqueued updates, in order: one D2H per ring (normally one per call), the only host sync."""
        raw = self._ring.read([(ref, row) for ref, row, _ in handles])   # (n, 160) bytes -> nested lists of Python floats:
        parts = (raw[:, :32].view(np.float64).tolist(), raw[:, 32:128].view(np.float64).tolist(),   # one conversion, not
                 raw[:, 128:136].view(np.float32).tolist(), raw[:, 136:148].view(np.float32).tolist())  # views per update
        return [self._info(part, h[2]) for part, h in zip(zip(*parts), handles)]

    def update(self, batch):
        return self.resolve([self.enqueue(batch)])[0]

    def _info(self, raw, B):
        algo, A = self.algo, self.A
        Bg = B * dist.world_size()                                       # the sums were reduced over all ranks
        sums, mom, aout, norms = raw                                     # 4 sums, 3 x (mean, std, max, min), (alpha, loss), 3 norms
        w_std, w_mean = algo.policy_std_reg_weight, algo.policy_mean_reg_weight
        reg = 0.0
        if w_std or w_mean:
            n = Bg * A - 1
            ms_ls = mom[1] ** 2 * n / (n + 1) + mom[0] ** 2                           # E[x^2] from mean / unbiased std
            ms_mu = mom[9] ** 2 * n / (n + 1) + mom[8] ** 2
            reg = w_std * ms_ls + w_mean * ms_mu
        info = {'Reward_Mean': sums[3] / Bg}
        if algo.automatic_entropy_tuning:
            info["Alpha"], info["Alpha_loss"] = aout
        info['Training/policy_loss'] = sums[2] / Bg + reg
        info['Training/qf1_loss'] = sums[0] / Bg
        info['Training/qf2_loss'] = sums[1] / Bg
        if algo.grad_clip is not None:
            info['Training/pf_grad_norm'], info['Training/qf1_grad_norm'], info['Training/qf2_grad_norm'] = norms
        for k, key in enumerate(("log_std", "log_probs", "mean")):
            info[key + '/mean'], info[key + '/std'], info[key + '/max'], info[key + '/min'] = mom[4 * k:4 * k + 4]
        return info
