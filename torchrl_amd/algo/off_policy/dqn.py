"""DQN and QR-DQN on the HIP path (torchrl/algo/off_policy/dqn.py:9-92, qrdqn.py:11-74).

`update(batch)`: Q(s) and Q'(s') through the conv + dense kernels (uint8 frame stacks are read
as stored and scaled in the first im2col), the TD / quantile-Huber loss kernel returns the loss
sums and dL/dQ, the backward pass fills one flat gradient buffer, trl_clip_adam_f32 steps it,
trl_polyak_f32 (soft) or a full copy every `target_hard_update_period` updates the target net.
Actions are read as (B,) or (B, 1) floats as the replay buffer stores them and cast to an index inside the
loss launch (the reference's two classes disagree on the shape, its Q16).
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


class DQN(OffRLAlgo):
    quantile_num = 1

    def __init__(self, qf, pf, qlr, optimizer_class=optim.Adam, optimizer_info={}, **kwargs):
        super().__init__(**kwargs)
        self.pf = pf
        self.qf = qf
        self.target_qf = copy.deepcopy(qf)
        self.qlr = qlr
        self.optimizer_class = optimizer_class
        self.optimizer_info = dict(optimizer_info)
        self.qf_optimizer = optimizer_class(self.qf.parameters(), lr=self.qlr, **optimizer_info)
        self.to(self.device)
        self._engine = None

    @property
    def networks(self):
        return [self.qf, self.target_qf]

    @property
    def target_networks(self):
        return [(self.qf, self.target_qf)]

    @property
    def snapshot_networks(self):
        return [("pf", self.qf)]

    def engine(self):
        if self._engine is None:
            if self.optimizer_class is not optim.Adam:
                raise _C.TrlError("the fused DQN step implements torch.optim.Adam only")
            self._engine = _FusedDQN(self)
        return self._engine

    def update(self, batch):
        self.training_update_num += 1
        return self.engine().update(batch)

    def static_batch(self):
        return self.engine().static_batch()

    def update_deferred(self, batch):
        """`update` without its read-back (see OffRLAlgo.update_per_epoch): returns a handle for `resolve_updates`."""
        self.training_update_num += 1
        return self.engine().enqueue(batch)

    def update_epoch_deferred(self, count):
        """All `count` {uniform sample -> update} pairs of an epoch as one replayed graph, or None (one by one then)."""
        handles = self.engine().enqueue_epoch(count)
        if handles is not None:
            self.training_update_num += count
        return handles

    def resolve_updates(self, handles):
        return self.engine().resolve(handles)


class QRDQN(DQN):
    def __init__(self, quantile_num=100, **kwargs):
        super().__init__(**kwargs)
        self.quantile_num = quantile_num


class _FusedDQN:
    def __init__(self, algo):
        self.algo = algo
        self.dev = next(algo.qf.parameters()).device
        if self.dev.type != "cuda":
            raise _C.TrlError("DQN networks live on %s: the HIP path needs a GPU (no CPU path exists)" % self.dev)
        from ...networks.base import MLPBase
        self.is_mlp = isinstance(algo.qf.base, MLPBase)                 # state-vector Q net (examples/dqn_state_vec.py)
        if self.is_mlp:
            self.layers, self.tlayers = ops.linear_layers(algo.qf), ops.linear_layers(algo.target_qf)
            self.act = ops.act_code(algo.qf)
            plist = [t for wb in self.layers for t in wb]
            tlist = [t for wb in self.tlayers for t in wb]
        else:
            plist, tlist = ops.cnn_param_list(algo.qf), ops.cnn_param_list(algo.target_qf)
        self.flat = flatten_into(plist)
        self.tflat = flatten_into(tlist)
        self.grads = torch.zeros_like(self.flat)
        self.m, self.v = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        self.gviews, off = [], 0
        for k in range(0, len(plist), 2):
            w, b = plist[k], plist[k + 1]
            gw = self.grads[off:off + w.numel()].view(w.shape); off += w.numel()
            gb = self.grads[off:off + b.numel()].view(b.shape); off += b.numel()
            self.gviews.append((gw, gb))
        off = 0
        for p in plist:
            n = p.numel()
            algo.qf_optimizer.state[p] = {"step": torch.tensor(0.0), "exp_avg": self.m[off:off + n].view(p.shape),
                                          "exp_avg_sq": self.v[off:off + n].view(p.shape)}
            off += n
        self.step_count = 0
        self.sums = torch.zeros(3, dtype=torch.float64, device=self.dev)
        self.A = int(algo.env.action_space.n)
        self.workspace = None
        # update u's three sums are filed into ring row u % slots by its loss launch (one rank), or copied there
        self._ring = StatRing(max(64, int(getattr(algo, "opt_times", 1))), 3, torch.float64, self.dev)
        self._slab = IndexSlab(self.dev)
        self._static, self._graphs, self._seen = None, {}, set()
        self.step_state = torch.tensor([0.0, 1.0, 1.0, 0.0], dtype=torch.float64, device=self.dev)
        self._head_ws = None                                            # workspace of the one-launch head (conv nets, Q = 1)

    def update(self, batch):
        return self.resolve([self.enqueue(batch)])[0]

    def resolve(self, handles):
        """Info dicts of enqueued updates, in order, after one D2H per ring of loss sums (the only host sync)."""
        rows = self._ring.read([(h[0], h[1]) for h in handles]).tolist()
        return [{'Reward_Mean': s[2] / B, 'Training/qf_loss': s[0] / denom, 'epsilon': eps, 'q_s_a': s[1] / (B * Q)}
                for s, (_, _, B, denom, Q, eps) in zip(rows, handles)]

    def static_batch(self):
        """Fixed-address input tensors of an update (None until the first batch has shown the shapes): the replay gather
        writes straight into them (`random_batch(..., out=)`) and the captured launch sequence reads them."""
        return self._static

    def _load(self, batch):
        dev = self.dev
        obs = batch['obs']
        if not self.is_mlp and not (isinstance(obs, torch.Tensor) and obs.dtype == torch.uint8):
            raise _C.TrlError("DQN.update expects uint8 (B, C, H, W) frame stacks from the device replay buffer")
        want = torch.uint8 if not self.is_mlp else torch.float32
        B = int(obs.shape[0])
        st = self._static
        if st is None or int(st["obs"].shape[0]) != B:
            shape = tuple(int(v) for v in (obs.shape if isinstance(obs, torch.Tensor) else np.asarray(obs).shape))
            f = lambda *sh: torch.zeros(sh, dtype=torch.float32, device=dev)
            st = {"obs": torch.zeros(shape, dtype=want, device=dev), "next_obs": torch.zeros(shape, dtype=want, device=dev),
                  "acts": f(B, 1), "rewards": f(B, 1), "terminals": f(B, 1)}
            self._static, self._graphs, self._seen = st, {}, set()
        for k in ("obs", "next_obs", "acts", "rewards", "terminals"):
            src = batch[k]
            if src is st[k]:
                continue
            src = src if isinstance(src, torch.Tensor) else torch.as_tensor(np.asarray(src))
            st[k].copy_(src.to(device=dev, dtype=st[k].dtype).reshape(st[k].shape), non_blocking=True)
        return st, B

    def _sequence(self, st, soft):
        """The fixed launch sequence of one update on the static inputs (eager, or under graph capture)."""
        algo, dev = self.algo, self.dev
        obs, nobs = st["obs"], st["next_obs"]
        rew, term = st["rewards"].view(-1), st["terminals"].view(-1)
        acts = st["acts"].view(-1)                                       # floats as stored; the loss launch casts at the read
        ring = (self._ring.t, self.step_state) if dist.world_size() == 1 else None    # (else: copied after the all-reduce)
        B, A, Q = int(obs.shape[0]), self.A, int(algo.quantile_num)
        pair = None
        if self.is_mlp:
            q, tape = ops.mlp_forward(self.layers, obs, self.act)
            qn, _ = ops.mlp_forward(self.tlayers, nobs, self.act)
        else:
            # online net on obs and target net on next_obs: one grouped launch per layer after the first
            (hw, hb), (tw, tb) = ops.fc_layers(algo.qf)[-1], ops.fc_layers(algo.target_qf)[-1]
            if Q == 1 and hb is not None and _C.dqn_head_supported(int(hw.shape[1]), A) and \
                    hw.data_ptr() % 16 == 0 and tw.data_ptr() % 16 == 0:
                pair = ops.cnn_forward_pair(algo.qf, algo.target_qf, obs, nobs, head=False, dx_prep=True)
            if pair is None:
                (q, tape), (qn, _) = ops.cnn_forward_pair(algo.qf, algo.target_qf, obs, nobs, dx_prep=True)
        if pair is not None:
            # the A-wide head: both forward passes, the loss and its whole backward pass in one launch
            (h, tape), (hn, _) = pair
            if self._head_ws is None:
                self._head_ws = _C.dqn_head_workspace(int(hw.shape[1]), A, dev)
            gw, gb = self.gviews[-1]
            dq = _C.dqn_head(h, hn, hw, hb, tw, tb, acts, rew, term, algo.discount, gw, gb, self.sums, self._head_ws,
                             ring=ring)                                  # = d loss / d h
        elif Q == 1:
            dq = _C.dqn_td_loss(q, acts, qn, rew, term, algo.discount, self.sums, ring=ring)
        else:
            dq = _C.quantile_huber(q, acts, qn, rew, term, algo.discount, A, Q, self.sums, ring=ring)
        if self.is_mlp:
            need = max(_C.lib().trl_linear_bwd_weight_workspace(B, int(w.shape[1]), int(w.shape[0])) for w, _ in self.layers)
        else:
            need = ops.cnn_backward_workspace(tape)                          # every conv layer its own region (one fold launch)
            need = max(need, max(_C.lib().trl_linear_bwd_weight_workspace(B, int(w.shape[1]), int(w.shape[0]))
                                 for w, _ in ops.fc_layers(algo.qf)))
        if self.workspace is None or self.workspace.numel() < need:
            self.workspace = torch.empty(need, device=dev)
        if self.is_mlp:
            ops.mlp_backward(tape, dq, grads=self.gviews, workspace=self.workspace)
        else:
            ops.cnn_backward(algo.qf, tape, dq, self.gviews, workspace=self.workspace)
        a = _C.AdamArgs()
        a.params, a.grads, a.exp_avg, a.exp_avg_sq = (self.flat.data_ptr(), self.grads.data_ptr(),
                                                      self.m.data_ptr(), self.v.data_ptr())
        a.n_groups = 1
        a.group_sizes[0] = self.flat.numel()
        a.group_lr[0] = algo.qf_optimizer.param_groups[0]['lr']
        a.max_norm, a.beta1, a.beta2 = 0.0, 0.9, 0.999
        a.eps = float(algo.optimizer_info.get("eps", 1e-8))
        dist.all_reduce_sum_(self.grads)                                 # C1: local-mean gradients -> SUM / world
        a.grad_scale, a.norms_out = 1.0 / dist.world_size(), None
        a.step_count, a.step_state = 0, self.step_state.data_ptr()       # the step count lives on the device
        if soft:                                                         # Adam, then Polyak (which also advances the step state)
            _C.clip_adam_polyak(a, self.tflat, self.flat, algo.tau, dev)
        else:
            _C.clip_adam(a, dev)
        dist.all_reduce_sum_(self.sums)

    def _run(self, st, soft):
        """Eager on the first visit of a configuration, captured into a HIP graph on the second, replayed afterwards
        (TRL_NO_GRAPH=1, or collectives between ranks: eager launches): the host then issues one call per update instead
        of ~45 and stays ahead of the device."""
        algo = self.algo
        key = (int(st["obs"].shape[0]), soft, float(algo.qf_optimizer.param_groups[0]['lr']), float(algo.discount),
               float(algo.tau), int(algo.quantile_num))
        if os.environ.get("TRL_NO_GRAPH") == "1" or dist.collectives_active() or \
                (key not in self._graphs and len(self._graphs) >= 4):
            self._sequence(st, soft)
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
        """Launch one update without waiting for it; its loss sums wait in their ring row for `resolve`."""
        algo = self.algo
        st, B = self._load(batch)
        soft = bool(algo.use_soft_update)
        self._ring.make_room(1)
        self._run(st, soft)
        self.step_count += 1
        if not soft and algo.training_update_num % algo.target_hard_update_period == 0:
            _C.polyak(self.tflat, self.flat, 1.0)
        handle = self._handles(1, B)[0]
        if dist.world_size() > 1:
            self._ring.t[handle[1]].copy_(self.sums, non_blocking=True)
        return handle

    def _handles(self, count, B):
        algo, world = self.algo, dist.world_size()
        Q = int(algo.quantile_num)
        denom = (float(B) if Q == 1 else float(B) * Q * Q) * world
        return [(ref, row, B * world, denom, Q, algo.pf.epsilon)
                for ref, row in self._ring.handles(self.step_count - count, count)]

    def enqueue_epoch(self, count):
        """`count` x {uniform replay sample -> update} as ONE graph launch (see _FusedSAC.enqueue_epoch): index sets drawn on
        the host in the reference's order and uploaded once, each update of the graph gathering the set the device-resident
        update count selects.  None when it does not apply: several ranks, a hard target copy due inside the epoch, a
        replay buffer that stores the sampled keys differently (frame-dedup), no update seen yet, TRL_NO_GRAPH=1."""
        algo = self.algo
        buf, B, st = getattr(algo, "replay_buffer", None), int(algo.batch_size), self._static
        soft = bool(algo.use_soft_update)
        period, done = int(algo.target_hard_update_period), int(algo.training_update_num)
        if buf is None or st is None or count < 1 or count > self._ring.slots or dist.world_size() != 1 or \
                dist.collectives_active() or os.environ.get("TRL_NO_GRAPH") == "1" or not hasattr(buf, "gather_sources") or \
                int(st["obs"].shape[0]) != B or (not soft and any((done + j) % period == 0 for j in range(1, count + 1))):
            return None
        keys = ("obs", "next_obs", "acts", "rewards", "terminals")
        srcs = buf.gather_sources(keys)
        nrows = buf._rows_per_batch(B)
        if srcs is None or any(s.dtype != st[k].dtype or s[0].numel() * nrows != st[k].numel() for s, k in zip(srcs, keys)):
            return None
        key = ("epoch", count, B, nrows, tuple(s.data_ptr() for s in srcs), soft,
               float(algo.qf_optimizer.param_groups[0]['lr']), float(algo.discount), float(algo.tau), int(algo.quantile_num))
        if key not in self._graphs and len(self._graphs) >= 4:
            return None
        self._ring.make_room(count)
        slab = self._slab.upload(self.step_count, buf.draw_indices(B, count))
        dsts = [st[k] for k in keys]

        def launches():
            for _ in range(count):
                _C.gather_rows_multi(srcs, slab, dsts, slab_counter=self.step_state, n_rows=nrows)
                self._sequence(st, soft)
        if key in self._graphs:
            self._graphs[key].replay()
        elif key not in self._seen:                                      # first visit eager, captured on the second
            self._seen.add(key)
            launches()
        else:
            graph, _ = _C.capture_graph(launches)
            self._graphs[key] = graph
            graph.replay()
        self.step_count += count
        return self._handles(count, B)
