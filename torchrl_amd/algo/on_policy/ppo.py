"""PPO on the fused HIP path (reference: torchrl/algo/on_policy/ppo.py:10-160).

Per epoch (`update_per_epoch`, ppo.py:27-39):
  1. last_value + GAE scan over the rollout (kernels, on_rl_algo.py:22-33);
  2. linear LR decay for both optimisers, `target_pf <- pf` (utils.py:23-32);
  3. `opt_epochs` permutations of the time rows from the global numpy RNG -- the
     reference's index stream, one permutation per pass (on_policy.py:76-78);
  4. advantage statistics of EVERY minibatch in one launch (trl_adv_stats_f64);
  5. per minibatch, three launches: fused gradient (critic + actor, fp32 MFMA,
     row gather fused in), partial fold, global-norm clip + Adam for both nets.
     Critic and actor are independent networks, so computing both gradients
     before either step is the same arithmetic as the reference's
     critic-then-actor order (ppo.py:149-150).
Nothing is read back until the epoch ends; then one copy fetches the statistics
of all minibatches and the logger receives the same info dicts, in order.

`update(batch)` keeps the reference's single-minibatch entry point (ppo.py:124-152).
"""
import copy
import os
import math

import numpy as np
import torch
import torch.optim as optim

from ... import _C
from ... import dist
from ...networks import flatten_into
from .. import utils as atu
from .a2c import A2C

_HALF_LOG_2PI_PLUS_HALF = 0.5 + 0.5 * math.log(2 * math.pi)


class PPO(A2C):
    loss_mode = _C.LOSS_PPO_CLIP

    def __init__(self, pf, clip_para=0.2, opt_epochs=10, clipped_value_loss=False, **kwargs):
        self.target_pf = copy.deepcopy(pf)
        super().__init__(pf=pf, **kwargs)
        self.clip_para = clip_para
        self.opt_epochs = opt_epochs
        self.clipped_value_loss = clipped_value_loss
        self.sample_key = ["obs", "acts", "advs", "estimate_returns", "values"]
        self._engine = None

    @property
    def networks(self):
        return [self.pf, self.vf, self.target_pf]

    def engine(self):
        if self._engine is None:
            if self.optimizer_class is not optim.Adam:
                raise _C.TrlError("the fused PPO step implements torch.optim.Adam only")
            self._engine = make_engine(self)
        return self._engine

    # ---- epoch ----
    def _fill_old_logp(self):
        """log pi_old of every stored (obs, act) under target_pf (ppo.py:54-56), computed once per
        epoch; update_per_epoch skips it when the collector kernel already wrote it with the same parameters."""
        buf = self.replay_buffer
        rows, n = buf._max_replay_buffer_size, buf.env_nums
        tgt = self.target_pf
        with torch.no_grad():                                          # kernels only: MLP forward + trl_gauss_logp_f32
            mean, _, log_std = tgt.forward(buf._obs.reshape(rows * n, -1))
            _C.gauss_logp(mean.contiguous(), buf._acts.reshape(rows * n, -1), log_std.float().contiguous(),
                          bool(tgt.tanh_action), out=buf._ensure_key("old_logp", (n, 1)).view(rows * n))

    def update_per_epoch(self):
        """ppo.py:27-39.  Everything the host decides -- the linear LR decay and the `opt_epochs` permutations of the
        time rows (global numpy stream, one per pass) -- is done first; everything on the device -- last value + GAE
        scan (on_rl_algo.py:22-33), `target_pf <- pf`, log pi_old, then all minibatch updates -- is handed to the engine
        as ONE launch sequence, which a single process replays as one HIP graph from its third visit on."""
        atu.update_linear_schedule(self.pf_optimizer, self.current_epoch, self.num_epochs, self.plr)
        atu.update_linear_schedule(self.vf_optimizer, self.current_epoch, self.num_epochs, self.vlr)
        buf = self.replay_buffer
        buf._scan_inputs()                                             # advs / estimate_returns exist before they are named
        buf._ensure_key("old_logp", (buf.env_nums, 1))
        passes = [buf.epoch_row_indices(self.batch_size, self.shuffle) for _ in range(self.opt_epochs)]
        row_idx = np.concatenate(passes, axis=0)                       # (E * n_mb, B // N)
        tensors = {"obs": buf._obs, "acts": buf._acts, "advs": buf._advs, "rets": buf._estimate_returns,
                   "old_values": buf._values, "old_logp": buf._old_logp}
        fresh = bool(getattr(buf, "_old_logp_fresh", False))          # the collector kernel wrote log pi_old already
        buf._old_logp_fresh = False

        def device_prologue():
            self.process_epoch_samples()
            self.engine().sync_target_pf(in_prologue=True)             # target_pf <- pf (utils.py:23-26)
            if not fresh:
                self._fill_old_logp()
        self._run_and_log(tensors, row_idx, buf.env_nums, pre=device_prologue,
                          pre_key=(fresh, self.gae, bool(getattr(buf, "_boot_fresh", False))))

    # ---- single minibatch (reference entry point) ----
    def update(self, batch):
        self.training_update_num += 1
        dev = self.device
        as_t = lambda x: (x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))) \
            .to(device=dev, dtype=torch.float32).contiguous()
        obs, acts = as_t(batch['obs']), as_t(batch['acts'])
        B = obs.shape[0]
        if "old_logp" in batch:
            old_lp = as_t(batch["old_logp"]).reshape(B, 1)
        else:
            with torch.no_grad():
                old_lp = self.target_pf.update(obs, acts)["log_prob"].reshape(B, 1).contiguous()
        tensors = {"obs": obs.reshape(1, B, -1), "acts": acts.reshape(1, B, -1),
                   "advs": as_t(batch['advs']).reshape(1, B, 1), "rets": as_t(batch['estimate_returns']).reshape(1, B, 1),
                   "old_values": as_t(batch['values']).reshape(1, B, 1), "old_logp": old_lp.reshape(1, B, 1)}
        return self.engine().run(tensors, np.zeros((1, 1), dtype=np.int64), B)[0]


class _FusedPPO:
    """Flat parameter / optimiser-state buffers and the per-minibatch launch sequence."""

    def __init__(self, algo):
        self.algo = algo
        pf, vf = algo.pf, algo.vf
        ps, vs = pf.mlp2_spec(), vf.mlp2_spec()
        if ps is None or vs is None or not hasattr(pf, "logstd"):
            raise _C.TrlError("fused PPO needs MLP2 nets and a GuassianContPolicyBasicBias policy")
        if vs[0] != ps[0] or vs[1] != ps[1] or vs[2] != 1 or vs[3] != ps[3]:
            raise _C.TrlError("policy %s and value %s must share input, width and activation" % (ps, vs))
        self.D, self.H, self.A, self.act = ps
        self.dev = next(pf.parameters()).device
        if self.dev.type != "cuda":
            raise _C.TrlError("PPO networks live on %s: the fused path needs a GPU (no CPU path exists)" % self.dev)
        pf_list = pf._mlp2_param_list() + [pf.logstd]
        vf_list = vf._mlp2_param_list()
        self.P_pf = sum(p.numel() for p in pf_list)
        self.P_vf = sum(p.numel() for p in vf_list)
        self.flat = flatten_into(pf_list + vf_list)                   # [pf | vf], parameters become views
        pf._flat, vf._flat = self.flat[:self.P_pf], self.flat[self.P_pf:]
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.grads = torch.zeros_like(self.flat)
        self.step_count = 0
        # target_pf mirrors the policy block of `flat`: its parameters become views of one buffer too
        tgt = getattr(algo, "target_pf", None)
        self.target_flat = None
        if tgt is not None:
            self.target_flat = flatten_into(tgt._mlp2_param_list() + [tgt.logstd])
        self._alias_optimizer_state(algo.pf_optimizer, pf_list, 0)
        self._alias_optimizer_state(algo.vf_optimizer, vf_list, self.P_pf)
        self.p_stride = _C.ppo_partial_stride(self.D, self.H, self.A)
        self.n_cu = torch.cuda.get_device_properties(self.dev).multi_processor_count
        self.max_wg = 2 * max(1, self.n_cu // 2)
        self.partial = torch.zeros(self.max_wg, self.p_stride, device=self.dev)
        self.scal = torch.zeros(self.max_wg, 8, dtype=torch.float64, device=self.dev)
        n_ws = _C.lib().trl_ppo_step_workspace(self.D, self.H, self.A)   # (begins with trl_ppo_reduce_adam_f32's workspace)
        self.red_ws = torch.zeros(n_ws, device=self.dev)              # Adam header + flags / norm granules of the fused step
        # TRL_PPO_STEP=fused (opt-in): the whole minibatch step as ONE launch (trl_ppo_minibatch_step_f32 -- its workgroups
        # meet inside the launch, so the grid must be resident at once; one process per device).  Same bits as the default
        # two-launch sequence (gradient, then fold / clip / Adam), and on MI355X 4 us per step SLOWER: rows that cross XCDs
        # inside a launch must be written through to the coherence point and every dependent hop costs 2.2-2.6 us, which is
        # what a graph-captured launch boundary costs too (profiles/NOTES_r06.md).
        self.one_launch = os.environ.get("TRL_PPO_STEP", "split") == "fused"
        self.step_max_wg = _C.lib().trl_ppo_step_max_workgroups()
        # One process: the critic's and the actor's updates of an epoch (ppo.py:93-122 / 41-91: separate networks, optimisers,
        # clips and statistics) run as TWO launch sequences on two streams (`_run_chains`); TRL_PPO_CHAINS=joint keeps the
        # single sequence in which every gradient launch carries both networks.  Same bits either way.
        self.two_chains = os.environ.get("TRL_PPO_CHAINS", "two") != "joint"
        self.red_ws_v = torch.zeros(n_ws, device=self.dev)            # the value chain's own Adam header + norm granules
        self.red_ws_v[4:8].view(torch.float64).fill_(1.0)
        self._side, self._value_done, self._hdr_owner, self._chain_graphs = None, None, "joint", {}
        self._pending, self._chain_pend, self._chain_seen = None, [], set()         # deferred statistics; shapes already run eagerly
        self._stats2_key = self._chain_rows = self._pro_ws = None                    # (allocated by the first run of a shape)
        self.red_ws[4:8].view(torch.float64).fill_(1.0)               # beta1^0, beta2^0 (device-side Adam state)

    def _alias_optimizer_state(self, opt, plist, offset):
        self._opt_steps = getattr(self, "_opt_steps", [])
        for p in plist:
            n = p.numel()
            step = torch.tensor(0.0)
            opt.state[p] = {"step": step, "exp_avg": self.m[offset:offset + n].view(p.shape),
                            "exp_avg_sq": self.v[offset:offset + n].view(p.shape)}
            self._opt_steps.append(step)
            offset += n

    def sync_target_pf(self, in_prologue=False):
        """target_pf <- pf.  in_prologue (called from an epoch's device prologue): the copy rides in the launch that
        computes the advantage statistics (`_C.ppo_epoch_prologue`) instead of being a launch of its own."""
        if self.target_flat is not None and in_prologue:
            self._copy_in_prologue = True
        elif self.target_flat is not None:
            self.target_flat.copy_(self.flat[:self.P_pf])
        else:
            atu.copy_model_params_from_to(self.algo.pf, self.algo.target_pf)

    def _n_wg(self, n_samples):
        """(workgroups to launch, how many of them run the policy) for one minibatch."""
        tiles = (n_samples + 15) // 16
        n_wg = 2 * max(1, min(self.max_wg // 2, (tiles + 3) // 4))
        n_pf = _C.lib().trl_ppo_wg_split(self.D, self.H, self.A, tiles, n_wg)
        if not 0 < n_pf < n_wg:
            raise _C.TrlError("trl_ppo_wg_split(%d tiles, %d workgroups) returned %d" % (tiles, n_wg, n_pf))
        if n_wg >= self.n_cu and n_wg % 8 == 0 and n_wg >= 16:
            # A full-chip grid: workgroup i runs on XCD i % 8, one workgroup per CU.  When the two networks' workgroups are
            # launched as SEPARATE kernels (two chains), a split that is not a multiple of 8 puts 33 workgroups on some XCDs
            # and 31 on others (147 + 109: 19 + 14 on XCDs 0-2) -- the 33rd waits a whole pass for a CU (round 6: 4.5-6 us
            # per launch).  The neighbouring multiples of 8 are compared with the cost model of trl_ppo_wg_split (a tile of
            # the policy costs 1.3 of the value net's; both counts of tiles per wave are what matters): 152 + 104 here.
            waves = lambda x: -(-tiles // (4 * x))
            cost = lambda x: max(waves(x) * 15.3, waves(n_wg - x) * 11.8)
            cands = [x for x in (n_pf // 8 * 8, -(-n_pf // 8) * 8) if 0 < x < n_wg]
            if cands:
                n_pf = min(cands, key=lambda x: (cost(x), abs(x - n_pf)))
        return n_wg, n_pf

    def _buffers(self, K, rows_mb):
        """Persistent per-shape buffers: minibatch row indices and the statistics block
        raw (K,4) f64 | info (K,24) f64 | norms (K,2) f32 on the device (stable addresses, so that a captured launch
        sequence can be replayed), plus their page-locked host twins: the per-epoch H2D of the indices and D2H of the
        statistics are asynchronous copies, the only host wait of an update is the one at its end."""
        key = (K, rows_mb)
        if getattr(self, "_buf_key", None) != key:
            self._buf_key = key
            self._idx_buf = torch.zeros(K * rows_mb, dtype=torch.int64, device=self.dev)
            self._stats = torch.zeros(29 * K, dtype=torch.float64, device=self.dev)
            # page-locked slab: the epoch's row indices, then one 8-byte slot for the two learning rates
            self._slab_host = torch.zeros(K * rows_mb + 1, dtype=torch.int64).pin_memory()
            self._idx_host = self._slab_host[:K * rows_mb]
            self._hyper_host = self._slab_host[K * rows_mb:].view(torch.float32)
            self._hyper = None
            self._stats_host = torch.zeros(29 * K, dtype=torch.float64).pin_memory()
            self._graph = None
        return self._idx_buf, self._stats

    def _set_device_hyper(self, lr_pf, lr_vf, upload=True):
        """The learning rates into their page-locked slot; upload=False: the epoch prologue launch copies them (and the
        row indices) itself, reading the slab in place."""
        hyper = (float(lr_pf), float(lr_vf))
        if getattr(self, "_hyper", None) != hyper or not upload:
            self._hyper = hyper if upload else None
            self._hyper_host[0], self._hyper_host[1] = hyper
            if upload:
                self.red_ws[2:4].copy_(self._hyper_host, non_blocking=True)

    defers = True                                                      # run(..., defer=True) is implemented

    def run(self, t, row_idx, N, pre=None, pre_key=None, defer=False):
        """t: dict of (rows, N, feat) device tensors; row_idx: (K, rows_mb) host int64.
        Runs K minibatch updates back to back; returns K info dicts (one host sync at the end).
        `defer=True`: returns a `_PendingInfos` right after the launches instead -- the wait for the statistics and the
        assembly of the K dicts happen when somebody asks for them (the logger at its next row) or at the start of the
        next `run`, by which time the next rollout is already running on the device: the host never idles the GPU
        between two iterations.
        `pre` (optional): a callable that enqueues device work which must precede the updates (PPO: last value + GAE
        scan, target_pf copy); it becomes part of the same launch sequence.
        Single process: the whole sequence -- `pre`, the statistics memset, the advantage statistics and the
        K x {gradient kernel, fused reduce/clip/Adam} launches -- is captured into a HIP graph the second time a shape
        is seen and replayed afterwards (dependent launches start ~2.5 us earlier each inside a graph on this machine,
        and the host issues one call instead of ~90); the Adam step count and learning rates live on the device so no
        launch argument changes between replays."""
        algo, dev = self.algo, self.dev
        K, rows_mb = row_idx.shape
        world = dist.world_size()
        n_local = rows_mb * N
        n_global = float(n_local * world)
        n_wg, n_wg_pf = self._n_wg(n_local)
        loss_mode = int(getattr(algo, "loss_mode", _C.LOSS_PPO_CLIP))
        probe = getattr(self, "probe", None)                           # bench.py: HIP events around the grad kernel
        fused = not dist.collectives_active()
        one_launch = fused and self.one_launch and n_wg <= self.step_max_wg
        # (env shards on several ranks: the two chains need the in-launch gradient exchange of the peer transport, whose
        # granules and exchange counts are per network; over all-reduce CALLS the joint sequence stays)
        xrank_chains = not fused and self.chains_across_ranks()
        if (fused or xrank_chains) and not one_launch and probe is None and self.two_chains and n_wg_pf >= 1 and n_wg - n_wg_pf >= 1:
            return self._run_chains(t, row_idx, N, pre, pre_key, defer, n_wg, n_wg_pf, loss_mode, n_global, xrank=not fused)
        for last in [self._pending] + list(self._chain_pend):
            if last is not None:                                       # its statistics still sit in the host twin this
                last.land()                                            # run is about to reuse: wait + snapshot (the dicts are
        self._pending, self._chain_pend = None, []                     # assembled when somebody reads them)
        idx_dev, stats = self._buffers(K, rows_mb)
        self._idx_host.numpy()[:] = row_idx.reshape(-1)
        rows_total = t["advs"].shape[0]
        raw, info = stats[:4 * K].view(K, 4), stats[4 * K:28 * K].view(K, 24)
        norms = stats[28 * K:].view(torch.float32).view(K, 2)
        # Env shards on several ranks with the peer transport up (dist.init_comm): the gradient SUM over ranks happens
        # INSIDE the fold / clip / Adam launch (trl_ppo_reduce_adam_xrank_f32) and the statistics go through the
        # one-kernel all-reduce -- plain launches, so the sequence is graph-replayed exactly like the single-process one.
        xrank = not fused and dist.peer_ready()
        self._settle_value_chain()                                     # (a joint launch sequence after a two-chain one)
        if self._value_done is not None:
            from ...networks import nets as _nets
            _nets.set_pending(algo.vf, None)
            self._value_done = None
        self._hdr_owner = "joint"                                      # (its Adam header is red_ws, which the policy chain kept current)
        # TRL_GRAPH_COLLECTIVES=1 (opt-in, RCCL route): capture the multi-rank sequence -- RCCL all-reduces included -- into
        # the HIP graph as well; the Adam step count and learning rates then live on the device like in the fused launch.
        graph_coll = not fused and not xrank and os.environ.get("TRL_GRAPH_COLLECTIVES") == "1"
        lr_pf, lr_vf = algo.pf_optimizer.param_groups[0]['lr'], algo.vf_optimizer.param_groups[0]['lr']
        # One process: no copy command for the epoch's row indices and learning rates -- the prologue launch reads the
        # page-locked slab in place and leaves the device copies the update kernels use.  (The slab is rewritten only after
        # the previous run's statistics have landed, see `last.resolve()` above: that run's launches are done by then.)
        in_place = fused and probe is None
        if not in_place:
            idx_dev.copy_(self._idx_host, non_blocking=True)
        if fused or xrank:
            self._set_device_hyper(lr_pf, lr_vf, upload=not in_place)
        elif graph_coll:
            if getattr(self, "step_state", None) is None:
                n = float(self.step_count)
                self.step_state = torch.tensor([n, 0.9 ** n, 0.999 ** n, 0.0], dtype=torch.float64, device=dev)
                self.lr_dev = torch.zeros(2, device=dev)
            if getattr(self, "_lr_host", None) != (float(lr_pf), float(lr_vf)):
                self._lr_host = (float(lr_pf), float(lr_vf))
                self.lr_dev.copy_(torch.tensor(self._lr_host, dtype=torch.float32), non_blocking=True)
        hyper = (float(getattr(algo, "clip_para", 0.0)), float(algo.entropy_coeff),
                 int(bool(getattr(algo, "clipped_value_loss", False))), int(bool(algo.pf.tanh_action)))
        use_graph = (fused or xrank or graph_coll) and probe is None and os.environ.get("TRL_NO_GRAPH") != "1"
        key = (n_wg, n_wg_pf, loss_mode, n_global, rows_total, N, pre_key, pre is not None, xrank, in_place, one_launch) + hyper + tuple(
            0 if t.get(k) is None else t[k].data_ptr() for k in ("obs", "acts", "advs", "rets", "old_values", "old_logp"))

        def launch_all():
            import ctypes as C
            g = _C.PpoBatchArgs()
            for k in ("obs", "acts", "advs", "rets", "old_values", "old_logp"):
                setattr(g, k, _C.dev_ptr(t[k], name=k).value if t.get(k) is not None else None)
            g.loss_mode = loss_mode
            g.rows_mb, g.N, g.n_global = rows_mb, N, n_global
            g.pf_params, g.vf_params = self.flat.data_ptr(), self.flat.data_ptr() + 4 * self.P_pf
            g.D, g.H, g.A, g.act = self.D, self.H, self.A, self.act
            g.clip_para, g.entropy_coeff, g.clipped_value_loss, g.tanh_action = hyper
            g.partial, g.scal_partial, g.n_wg, g.n_wg_pf = self.partial.data_ptr(), self.scal.data_ptr(), n_wg, n_wg_pf
            a = _C.AdamArgs()
            a.params, a.grads, a.exp_avg, a.exp_avg_sq = (self.flat.data_ptr(), self.grads.data_ptr(),
                                                          self.m.data_ptr(), self.v.data_ptr())
            a.n_groups = 2
            a.group_sizes[0], a.group_sizes[1] = self.P_pf, self.P_vf
            a.group_lr[0], a.group_lr[1] = lr_pf, lr_vf
            a.max_norm, a.beta1, a.beta2, a.eps, a.grad_scale = 0.5, 0.9, 0.999, 1e-5, 1.0
            a.device_state = int(fused or xrank)                       # step count / lr from the workspace header
            idx_base, raw_base, info_base, norm_base = idx_dev.data_ptr(), raw.data_ptr(), info.data_ptr(), norms.data_ptr()
            lib = _C.lib()
            self._copy_in_prologue = False
            if pre is not None:
                pre()
            stream = _C.stream_ptr(dev)
            # advantage statistics of all K minibatches; the statistics block cleared and target_pf <- pf in the same launch
            if self._pro_ws is None or self._pro_ws_k != (K, rows_mb):   # (its counters assume one slicing)
                self._pro_ws, self._pro_ws_k = _C.ppo_epoch_prologue_workspace(K, dev), (K, rows_mb)
            copies = [(self.target_flat, self.flat[:self.P_pf])] if self._copy_in_prologue else []
            if in_place:
                copies += [(idx_dev, self._idx_host), (self.red_ws[2:4], self._hyper_host)]
            _C.ppo_epoch_prologue(t["advs"].reshape(rows_total, N),
                                  (self._idx_host if in_place else idx_dev).view(K, rows_mb), raw, self._pro_ws,
                                  zero=stats[4 * K:], copies=copies)
            dist.reduce_adv_raw_(raw)
            for k in range(K):
                g.row_idx = idx_base + 8 * rows_mb * k
                g.adv_raw = raw_base + 32 * k
                a.step_count = self.step_count + k + 1
                a.norms_out = norm_base + 8 * k
                if probe is not None:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record()
                if one_launch:                                         # gradient + fold + clip + Adam: one launch
                    _C.check(lib.trl_ppo_minibatch_step_f32(C.byref(g), self.grads.data_ptr(), info_base + 192 * k,
                                                            C.byref(a), self.red_ws.data_ptr(), stream),
                             "trl_ppo_minibatch_step_f32")
                    if probe is not None:
                        ev[1].record()
                        probe.append(ev)
                    continue
                _C.check(lib.trl_ppo_minibatch_grad_f32(C.byref(g), stream), "trl_ppo_minibatch_grad_f32")
                if probe is not None:
                    ev[1].record()
                    probe.append(ev)
                if fused:                                              # one process: reduce + clip + Adam in one launch
                    _C.check(lib.trl_ppo_reduce_adam_f32(self.partial.data_ptr(), self.scal.data_ptr(), n_wg, n_wg_pf,
                                                         self.D, self.H, self.A, self.grads.data_ptr(), info_base + 192 * k,
                                                         C.byref(a), self.red_ws.data_ptr(), stream),
                             "trl_ppo_reduce_adam_f32")
                    continue
                if xrank:                                              # the same launch with the cross-rank SUM inside
                    _C.check(lib.trl_ppo_reduce_adam_xrank_f32(self.partial.data_ptr(), self.scal.data_ptr(), n_wg, n_wg_pf,
                                                               self.D, self.H, self.A, self.grads.data_ptr(),
                                                               info_base + 192 * k, C.byref(a), self.red_ws.data_ptr(),
                                                               dist.comm_handle(), stream), "trl_ppo_reduce_adam_xrank_f32")
                    continue
                _C.check(lib.trl_ppo_reduce_f32(self.partial.data_ptr(), self.scal.data_ptr(), n_wg, n_wg_pf, self.D,
                                                self.H, self.A, self.flat.data_ptr(), self.grads.data_ptr(),
                                                info_base + 192 * k, stream), "trl_ppo_reduce_f32")
                dist.all_reduce_sum_(self.grads)                       # C1: gradient SUM over ranks
                if graph_coll:
                    a.step_count, a.step_state, a.device_lr = 0, self.step_state.data_ptr(), self.lr_dev.data_ptr()
                _C.check(lib.trl_clip_adam_f32(C.byref(a), stream), "trl_clip_adam_f32")

        if not use_graph:
            launch_all()
        elif getattr(self, "_graph", None) is not None and self._graph_key == key:
            self._graph.replay()
        elif getattr(self, "_graph_seen", None) != key:
            self._graph_seen = key                                     # first visit of a shape: run eagerly (warm-up)
            self._graph = None
            launch_all()
        else:
            graph, _ = _C.capture_graph(launch_all)
            self._graph, self._graph_key = graph, key
            graph.replay()
        self.step_count += K
        dist.reduce_info_(info)
        self._stats_host.copy_(stats, non_blocking=True)
        landed = torch.cuda.Event()
        landed.record()
        for s in self._opt_steps:                                      # host bookkeeping under the device's shadow
            s.fill_(float(self.step_count))
        host = self._stats_host
        make = self._infos_a2c if loss_mode == _C.LOSS_A2C else self._infos

        def build(host):
            if xrank:
                dist.check_comm(peek=True)                             # a rank that never delivered: raise, do not hang
            if one_launch and host[4 * K:28 * K].view(K, 24)[:, 23].any():
                raise _C.TrlError("trl_ppo_minibatch_step_f32: the in-launch rendezvous of a minibatch step timed out (its "
                                  "workgroups were not resident together -- is another process using this GPU?); the "
                                  "parameters of that step were left untouched.  TRL_PPO_STEP=split uses two launches.")
            return make(host[:4 * K].view(K, 4).numpy(), host[4 * K:28 * K].view(K, 24).numpy(),
                        host[28 * K:].view(torch.float32).view(K, 2).numpy(), n_global)
        pending = _PendingInfos(K, landed, host, build)
        if defer:
            self._pending = pending
            return pending
        return pending.resolve()                                       # the only host wait of the update

    def chains_across_ranks(self):
        """Env shards on several ranks: do the updates run as two chains?  Needs the peer transport (the in-launch exchange
        is per network).  Default: yes with one rank per GPU -- each device then runs what a single process runs, plus the
        waits; no when ranks SHARE a device (tests, bench.py's device map): four or more chains and as many waiting fold
        launches on one chip form convoys (two ranks on one MI355X: 5.9 ms per iteration against 5.5 joint, round 6).
        TRL_PPO_CHAINS_XRANK=1 / 0 forces either."""
        if not (self.two_chains and dist.collectives_active() and dist.peer_ready()):
            return False
        want = os.environ.get("TRL_PPO_CHAINS_XRANK")
        return want == "1" if want in ("0", "1") else dist.ranks_per_device() == 1

    def _settle_value_chain(self):
        """The current stream waits for the value chain of the last two-chain run (a no-op when it was waited for already,
        e.g. by the value pass of a fused rollout)."""
        if self._value_done is not None:
            torch.cuda.current_stream(self.dev).wait_event(self._value_done)

    def _sync_headers(self, owner):
        """The joint sequence keeps its Adam header (step count, beta powers, learning rates) in `red_ws`; of the two chains
        the policy's uses `red_ws` and the value's `red_ws_v`.  Both chains step once per update, so the headers agree
        after every run -- a change of route just hands the current one over."""
        if self._hdr_owner != owner:
            if owner == "two":
                self.red_ws_v[:8].copy_(self.red_ws[:8])
            self._hdr_owner = owner

    def _run_chains(self, t, row_idx, N, pre, pre_key, defer, n_wg, n_wg_pf, loss_mode, n_global, xrank=False):
        """K minibatch updates as TWO launch sequences (one process): after the epoch's prologue on the current stream
        (`pre`, advantage statistics, target copy, uploads) the POLICY chain -- K x {gradient launch of the policy's
        workgroups, its fold / clip / Adam} -- stays on the current stream and the VALUE chain -- the same for the value
        function's workgroups -- goes to a side stream.  Critic and actor updates are independent given the epoch's
        advantages (ppo.py:93-122 / 41-91), so the results are those of the joint sequence, bit for bit (same tiles per
        workgroup, same fold order, same Adam arithmetic) -- but the NEXT ROLLOUT, which reads only the policy, starts as
        soon as the policy chain is through, on the CUs that chain frees, while the value chain (whose ten-tile
        workgroups make it the longer one) is still stepping: the rollout's 0.2 ms latency chain disappears under it.
        The value pass behind that rollout waits for the value chain's end event (trl_rollout_t.value_wait_event); every
        other reader of the value function's parameters settles through networks.nets.settle.
        `xrank` (env shards on several ranks, peer transport): each chain's fold launch carries its network's gradient SUM
        over ranks (trl_ppo_reduce_adam_xrank_net_f32: own exchange count, own granules); the advantage statistics are
        reduced in the head, the logged statistics of both chains behind the value chain, on its stream."""
        import ctypes as C
        algo, dev = self.algo, self.dev
        K, rows_mb = row_idx.shape
        main = torch.cuda.current_stream(dev)
        if self._side is None:
            self._side = torch.cuda.Stream(dev)
        side = self._side
        # The host may be ONE run ahead of the device: this run's launches are issued while the previous run's value chain
        # (which ends about when the rollout between the two does) may still be running -- waiting for it here would leave
        # the device waiting for the host right after the value pass.  Everything the host writes per run therefore exists
        # TWICE (page-locked index / learning-rate slab, which the prologue reads in place, and the statistics' host twin),
        # used in turn; the run before the previous one has to have landed before its set is reused.
        last = self._pending                                           # (a joint run before this one)
        if last is not None:
            last.land()
            self._pending = None
        pend = self._chain_pend
        while len(pend) > (0 if os.environ.get("TRL_CHAIN_HOST_AHEAD") == "0" else 1):   # (=0: development A/B, wait for the previous run)
            pend.pop(0).land()
        self._settle_value_chain()                                     # device-side order (normally long satisfied: the value pass waited)
        self._sync_headers("two")
        idx_dev, _ = self._buffers(K, rows_mb)
        if self._stats2_key != (K, rows_mb):
            for q in pend:
                q.land()
            del pend[:]
            self._stats2_key = (K, rows_mb)
            self._stats2 = torch.zeros(2, 29 * K, dtype=torch.float64, device=dev)        # [policy chain | value chain]
            self._chain_host = [(torch.zeros(K * rows_mb + 1, dtype=torch.int64).pin_memory(),
                                 torch.zeros(2, 29 * K, dtype=torch.float64).pin_memory()) for _ in range(2)]
            self._chain_graphs, self._chain_turn = {}, 0
        stats2 = self._stats2
        turn = self._chain_turn = 1 - self._chain_turn
        slab, stats_host = self._chain_host[turn]
        idx_host, hyper_host = slab[:K * rows_mb], slab[K * rows_mb:].view(torch.float32)
        idx_host.numpy()[:] = row_idx.reshape(-1)
        rows_total = t["advs"].shape[0]
        raw = stats2[0, :4 * K].view(K, 4)
        lr_pf, lr_vf = algo.pf_optimizer.param_groups[0]['lr'], algo.vf_optimizer.param_groups[0]['lr']
        hyper_host[0], hyper_host[1] = float(lr_pf), float(lr_vf)
        self._hyper = None                                             # (the joint route uploads its own copy when it runs next)
        hyper = (float(getattr(algo, "clip_para", 0.0)), float(algo.entropy_coeff),
                 int(bool(getattr(algo, "clipped_value_loss", False))), int(bool(algo.pf.tanh_action)))
        n_wg_vf = n_wg - n_wg_pf
        if self._chain_rows is None:
            self._chain_rows = (torch.zeros(self.max_wg, self.p_stride, device=dev), torch.zeros(self.max_wg, 8, dtype=torch.float64, device=dev))
        partial_v, scal_v = self._chain_rows                           # (the policy chain uses self.partial / self.scal)
        shape_key = (n_wg, n_wg_pf, loss_mode, n_global, rows_total, N, pre_key, pre is not None, xrank) + hyper
        key = shape_key + (turn,) + tuple(
            0 if t.get(k) is None else t[k].data_ptr() for k in ("obs", "acts", "advs", "rets", "old_values", "old_logp"))

        def head():
            self._copy_in_prologue = False
            if pre is not None:
                pre()
            if self._pro_ws is None or self._pro_ws_k != (K, rows_mb):
                self._pro_ws, self._pro_ws_k = _C.ppo_epoch_prologue_workspace(K, dev), (K, rows_mb)
            copies = [(self.target_flat, self.flat[:self.P_pf])] if self._copy_in_prologue else []
            copies += [(idx_dev, idx_host), (self.red_ws[2:4], hyper_host), (self.red_ws_v[2:4], hyper_host)]
            _C.ppo_epoch_prologue(t["advs"].reshape(rows_total, N), idx_host.view(K, rows_mb), raw, self._pro_ws,
                                  zero=stats2.view(-1)[4 * K:], copies=copies)
            dist.reduce_adv_raw_(raw)                                  # C2 (identity in one process)

        def chain(net):
            lib = _C.lib()
            stream = _C.stream_ptr(dev)
            g = _C.PpoBatchArgs()
            for k in ("obs", "acts", "advs", "rets", "old_values", "old_logp"):
                setattr(g, k, _C.dev_ptr(t[k], name=k).value if t.get(k) is not None else None)
            g.loss_mode = loss_mode
            g.rows_mb, g.N, g.n_global = rows_mb, N, n_global
            g.pf_params, g.vf_params = self.flat.data_ptr(), self.flat.data_ptr() + 4 * self.P_pf
            g.D, g.H, g.A, g.act = self.D, self.H, self.A, self.act
            g.clip_para, g.entropy_coeff, g.clipped_value_loss, g.tanh_action = hyper
            part, scal, ws = (self.partial, self.scal, self.red_ws) if net == 0 else (partial_v, scal_v, self.red_ws_v)
            g.partial, g.scal_partial = part.data_ptr(), scal.data_ptr()
            g.n_wg, g.n_wg_pf = (n_wg_pf, n_wg_pf) if net == 0 else (n_wg_vf, -1)
            a = _C.AdamArgs()
            a.params, a.grads, a.exp_avg, a.exp_avg_sq = (self.flat.data_ptr(), self.grads.data_ptr(),
                                                          self.m.data_ptr(), self.v.data_ptr())
            a.n_groups = 2
            a.group_sizes[0], a.group_sizes[1] = self.P_pf, self.P_vf
            a.group_lr[0], a.group_lr[1] = lr_pf, lr_vf
            a.max_norm, a.beta1, a.beta2, a.eps, a.grad_scale = 0.5, 0.9, 0.999, 1e-5, 1.0
            a.device_state = 1
            st = stats2[net]
            info_base, norm_base = st[4 * K:].data_ptr(), st[28 * K:].data_ptr()
            for k in range(K):
                g.row_idx = idx_dev.data_ptr() + 8 * rows_mb * k
                g.adv_raw = raw.data_ptr() + 32 * k
                a.step_count, a.norms_out = self.step_count + k + 1, norm_base + 8 * k
                _C.check(lib.trl_ppo_minibatch_grad_f32(C.byref(g), stream), "trl_ppo_minibatch_grad_f32")
                if xrank:                                              # the network's gradient SUM over ranks inside the launch
                    _C.check(lib.trl_ppo_reduce_adam_xrank_net_f32(part.data_ptr(), scal.data_ptr(), g.n_wg, net, self.D, self.H,
                                                                   self.A, self.grads.data_ptr(), info_base + 192 * k, C.byref(a),
                                                                   ws.data_ptr(), dist.comm_handle(), stream),
                             "trl_ppo_reduce_adam_xrank_net_f32")
                    continue
                _C.check(lib.trl_ppo_reduce_adam_net_f32(part.data_ptr(), scal.data_ptr(), g.n_wg, net, self.D, self.H, self.A,
                                                         self.grads.data_ptr(), info_base + 192 * k, C.byref(a), ws.data_ptr(), stream),
                         "trl_ppo_reduce_adam_net_f32")

        # (graphs by key, a few of them: the stored observations alternate between the ring's tensor and its shadow, see
        # collector/on_policy.py::_launch, so a steady run replays TWO captured sets in turn)
        use_graph = os.environ.get("TRL_NO_GRAPH") != "1"
        cache = self._chain_graphs if isinstance(self._chain_graphs, dict) else {}
        self._chain_graphs = cache
        seen = self._chain_seen
        graphs = cache.get(key) if use_graph else None
        if use_graph and graphs is None and shape_key in seen and len(cache) < 8:
            # a shape's first run is eager (kernels load, attributes are set); every set of addresses after that is captured
            # on its first visit -- the two sets of a steady run are both replaying from the fourth iteration on
            g0, _ = _C.capture_graph(head)
            gp, _ = _C.capture_graph(lambda: chain(0))
            with torch.cuda.stream(side):
                gv, _ = _C.capture_graph(lambda: chain(1))
            graphs = cache[key] = (g0, gp, gv)
        elif graphs is None:
            seen.add(shape_key)                                        # first visit of a shape: eager (warm-up)
        run_head, run_p, run_v = (head, lambda: chain(0), lambda: chain(1)) if graphs is None else \
            (graphs[0].replay, graphs[1].replay, graphs[2].replay)
        only = os.environ.get("TRL_CHAIN_ONLY")                        # development aid (tools/time_chains.py): one chain alone
        if only == "v":
            run_p = lambda: None
        if only == "p":
            run_v = lambda: None
        run_head()
        forked = torch.cuda.Event()
        forked.record(main)
        run_p()                                                        # policy chain: the current stream (the next rollout follows it)
        if not xrank:
            stats_host[0].copy_(stats2[0], non_blocking=True)
        landed_p = torch.cuda.Event()
        landed_p.record(main)
        with torch.cuda.stream(side):                                  # value chain: beside it, and beside the next rollout
            side.wait_event(forked)
            if getattr(self, "_test_value_chain_delay", 0):            # tests: hold the value chain back (device spin) so that the
                torch.cuda._sleep(int(self._test_value_chain_delay))   # next rollout really runs beside / ahead of it
            run_v()
            if xrank:
                # C3 for both chains' statistics, here: the current stream goes on to the rollout without another
                # collective, and the next one it issues (the next head's C2) comes after its value pass has waited for
                # this stream -- every rank issues the small all-reduces in the same order
                side.wait_event(landed_p)
                for st_ in (stats2[0], stats2[1]):
                    dist.reduce_info_(st_[4 * K:28 * K].view(K, 24))
                stats_host[0].copy_(stats2[0], non_blocking=True)
            stats_host[1].copy_(stats2[1], non_blocking=True)
            done_v = torch.cuda.Event()
            done_v.record(side)
        self._value_done = done_v
        from ...networks import nets as _nets
        _nets.set_pending(algo.vf, done_v)
        self.step_count += K
        for s_ in self._opt_steps:                                      # host bookkeeping under the device's shadow
            s_.fill_(float(self.step_count))
        make = self._infos_a2c if loss_mode == _C.LOSS_A2C else self._infos

        class _Both:
            @staticmethod
            def synchronize():
                landed_p.synchronize()
                done_v.synchronize()

        def build(host):
            if xrank:
                dist.check_comm(peek=True)                             # a rank that never delivered: raise, do not hang
            hp, hv = host[0], host[1]
            info = hp[4 * K:28 * K].view(K, 24).clone()
            iv = hv[4 * K:28 * K].view(K, 24)
            for col in (7, 12, 13, 14, 15):                             # the value chain's entries of the statistics row
                info[:, col] = iv[:, col]
            norms = hp[28 * K:].view(torch.float32).view(K, 2).clone()
            norms[:, 1] = hv[28 * K:].view(torch.float32).view(K, 2)[:, 1]
            return make(hp[:4 * K].view(K, 4).numpy(), info.numpy(), norms.numpy(), n_global)
        pending = _PendingInfos(K, _Both, stats_host, build)
        if defer:
            pend.append(pending)
            return pending
        out = pending.resolve()
        self._settle_value_chain()                                     # a caller that reads in place gets settled parameters too
        return out

    def _infos_a2c(self, raw, info, norms, n):
        """The info dict of A2C.update (a2c.py:86-105); `std` is (B, A) there, each dim repeated B times."""
        out = []
        c_ent = float(self.algo.entropy_coeff)
        A = self.A
        for r, i, g in zip(raw, info, norms):
            v_var = max((i[13] - i[12] * i[12] / n) / (n - 1), 0.0)
            ent = A * _HALF_LOG_2PI_PLUS_HALF + A * i[8]
            std_ss = (i[17] ** 2) * (A - 1) if A > 1 else 0.0            # sum over dims of (std - mean)^2
            out.append({
                'Training/policy_loss': i[0] / n - c_ent * ent, 'Training/vf_loss': i[7] / n,
                'v_pred/mean': i[12] / n, 'v_pred/std': math.sqrt(v_var), 'v_pred/max': i[14], 'v_pred/min': -i[15],
                'std/mean': i[16], 'std/std': math.sqrt(n * std_ss / (n * A - 1)), 'std/max': i[18], 'std/min': i[19],
                'ent': ent, 'log_prob': i[1] / n,
            })
        return out

    def _infos(self, raw, info, norms, n):
        """K info dicts with the keys of PPO.update (ppo.py:76-90, 120-122, 141-146), assembled column-wise."""
        c_ent = float(self.algo.entropy_coeff)
        r, i = raw, info
        adv_var = np.maximum((r[:, 1] - r[:, 0] * r[:, 0] / n) / (n - 1), 0.0)
        lp_var = np.maximum((i[:, 2] - i[:, 1] * i[:, 1] / n) / (n - 1), 0.0)
        ent = self.A * _HALF_LOG_2PI_PLUS_HALF + self.A * i[:, 8]
        cols = (('advs/mean', r[:, 0] / n), ('advs/std', np.sqrt(adv_var)), ('advs/max', r[:, 2]), ('advs/min', -r[:, 3]),
                ('Training/vf_loss', i[:, 7] / n), ('grad_norm/vf', norms[:, 1].astype(np.float64)),
                ('Training/policy_loss', i[:, 0] / n - c_ent * ent),
                ('logprob/mean', i[:, 1] / n), ('logprob/std', np.sqrt(lp_var)), ('logprob/max', i[:, 3]),
                ('logprob/min', -i[:, 4]), ('log_std/mean', i[:, 8]), ('log_std/std', i[:, 9]), ('log_std/max', i[:, 10]),
                ('log_std/min', i[:, 11]), ('ratio/max', i[:, 5]), ('ratio/min', -i[:, 6]),
                ('grad_norm/pf', norms[:, 0].astype(np.float64)))
        keys = [k for k, _ in cols]
        table = np.stack([v for _, v in cols], axis=1).tolist()
        return [dict(zip(keys, row)) for row in table]


class _PendingInfos:
    """The K info dicts of a launched update sequence: `resolve()` waits for the statistics' D2H copy (an event, not the
    whole stream) and assembles them, once.  `land()` is the wait alone plus a snapshot of the page-locked block: the
    next run calls it before it reuses that block, and leaves the assembly of the dicts (~0.1 ms of host time) to whoever
    reads them -- after the next launch sequence is on its way instead of in front of it."""

    def __init__(self, count, landed, host, build):
        self.count, self._landed, self._host, self._build, self._snap, self._infos = count, landed, host, build, None, None

    def __len__(self):
        return self.count

    def land(self):
        if self._snap is None and self._infos is None:
            self._landed.synchronize()
            self._snap, self._host = self._host.clone(), None

    def resolve(self):
        if self._infos is None:
            self.land()
            self._infos, self._build, self._snap = self._build(self._snap), None, None
        return self._infos


def make_engine(algo):
    """The fused engine when the networks have the shape its kernels are instantiated for, the generic one otherwise."""
    pf, vf = algo.pf, algo.vf
    ps = pf.mlp2_spec() if hasattr(pf, "mlp2_spec") else None
    vs = vf.mlp2_spec() if hasattr(vf, "mlp2_spec") else None
    if ps is not None and vs is not None and hasattr(pf, "logstd") and _C.lib().trl_ppo_partial_stride(ps[0], ps[1], ps[2]) > 0 \
            and os.environ.get("TRL_GENERIC_PPO") != "1":
        return _FusedPPO(algo)
    return _GenericPPO(algo)


class _GenericPPO(_FusedPPO):
    defers = False                                                     # (its run returns the dicts)
    """PPO / A2C minibatch loop for ARBITRARY MLP shapes (any observation / action size, width, depth): the layers run
    on the generic dense-layer kernels (k_gemm.hip, through ops.mlp_forward / mlp_backward), the loss half on
    trl_ppo_generic_losses_f32, clip + Adam on trl_clip_adam_f32.  Same interface, statistics block and info dicts
    as the fused engine, same per-sample arithmetic; ~25 launches per minibatch instead of 2 (TRL_GENERIC_PPO=1
    forces this engine for the benchmark shape too, which is how it is tested against the fused one)."""

    def sync_target_pf(self, in_prologue=False):
        super().sync_target_pf(False)                                  # (this engine's launch sequence has no fused prologue)

    def __init__(self, algo):
        from ... import ops
        self.algo, self.ops = algo, ops
        pf, vf = algo.pf, algo.vf
        if not hasattr(pf, "logstd"):
            raise _C.TrlError("PPO / A2C kernels need a state-independent-std policy (GuassianContPolicyBasicBias)")
        self.dev = next(pf.parameters()).device
        if self.dev.type != "cuda":
            raise _C.TrlError("PPO networks live on %s: the HIP path needs a GPU (no CPU path exists)" % self.dev)
        self.act = ops.act_code(pf)
        if ops.act_code(vf) != self.act:
            raise _C.TrlError("policy and value network must use the same activation")
        self.pf_layers, self.vf_layers = ops.linear_layers(pf), ops.linear_layers(vf)
        pf_list = [t for wb in self.pf_layers for t in wb] + [pf.logstd]
        vf_list = [t for wb in self.vf_layers for t in wb]
        self.P_pf = sum(p.numel() for p in pf_list)
        self.P_vf = sum(p.numel() for p in vf_list)
        self.D, self.A = int(self.pf_layers[0][0].shape[1]), int(pf.logstd.numel())
        self.flat = flatten_into(pf_list + vf_list)                   # [pf | vf], parameters become views
        # nets that are MLP2 blocks hand their own flat view to the fused inference kernel (Net.flat_params): it must
        # be THIS storage, or the first forward after the engine exists would re-home the parameters away from it
        for net, lo, hi in ((pf, 0, self.P_pf), (vf, self.P_pf, self.P_pf + self.P_vf)):
            if getattr(net, "mlp2_spec", lambda: None)() is not None:
                net._flat = self.flat[lo:hi]
        self.m, self.v, self.grads = torch.zeros_like(self.flat), torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        self.step_count = 0
        tgt = getattr(algo, "target_pf", None)
        self.target_flat = None
        if tgt is not None:
            self.target_flat = flatten_into([t for wb in ops.linear_layers(tgt) for t in wb] + [tgt.logstd])
        self._alias_optimizer_state(algo.pf_optimizer, pf_list, 0)
        self._alias_optimizer_state(algo.vf_optimizer, vf_list, self.P_pf)
        self.gviews, off = [], 0
        for layers in (self.pf_layers, self.vf_layers):
            views = []
            for w, b in layers:
                gw = self.grads[off:off + w.numel()].view(w.shape); off += w.numel()
                gb = self.grads[off:off + b.numel()].view(b.shape); off += b.numel()
                views.append((gw, gb))
            self.gviews.append(views)
            if layers is self.pf_layers:
                self.g_logstd = self.grads[off:off + self.A]; off += self.A
        self.step_state = torch.tensor([0.0, 1.0, 1.0, 0.0], dtype=torch.float64, device=self.dev)
        self.lr_dev = torch.zeros(2, device=self.dev)
        self._graphs, self._seen = {}, set()
        self.workspace = None

    def _ws(self, B):
        need = max(_C.lib().trl_linear_bwd_weight_workspace(B, int(w.shape[1]), int(w.shape[0]))
                   for w, _ in self.pf_layers + self.vf_layers)
        if self.workspace is None or self.workspace.numel() < need:
            self.workspace = torch.empty(need, device=self.dev)
        return self.workspace

    def run(self, t, row_idx, N, pre=None, pre_key=None):
        algo, dev, ops = self.algo, self.dev, self.ops
        K, rows_mb = row_idx.shape
        world = dist.world_size()
        n_local = rows_mb * N
        n_global = float(n_local * world)
        idx_dev, stats = self._buffers(K, rows_mb)
        idx_dev.copy_(torch.from_numpy(np.ascontiguousarray(row_idx).reshape(-1)), non_blocking=True)
        rows_total = t["advs"].shape[0]
        raw, info = stats[:4 * K].view(K, 4), stats[4 * K:28 * K].view(K, 24)
        norms = stats[28 * K:].view(torch.float32).view(K, 2)
        loss_mode = int(getattr(algo, "loss_mode", _C.LOSS_PPO_CLIP))
        ws = self._ws(n_local)
        idx2d = idx_dev.view(K, rows_mb)
        replayable = not dist.collectives_active() and os.environ.get("TRL_NO_GRAPH") != "1"
        lrs = (float(algo.pf_optimizer.param_groups[0]['lr']), float(algo.vf_optimizer.param_groups[0]['lr']))
        if getattr(self, "_lr_host", None) != lrs:                     # learning rates live on the device: the linear
            self._lr_host = lrs                                        # schedule changes no launch argument
            self.lr_dev.copy_(torch.tensor(lrs, dtype=torch.float32), non_blocking=True)
        gather = lambda key, k: None if t.get(key) is None else \
            _C.gather_rows(t[key].reshape(rows_total, N, -1), idx2d[k]).reshape(n_local, -1)
        hyper = (float(getattr(algo, "clip_para", 0.0)), float(algo.entropy_coeff),
                 bool(getattr(algo, "clipped_value_loss", False)), bool(algo.pf.tanh_action), loss_mode)

        def launch_all():
            if pre is not None:
                pre()
            stats.zero_()
            _C.adv_stats(t["advs"].reshape(rows_total, N), idx2d, raw)
            dist.reduce_adv_raw_(raw)
            for k in range(K):
                obs, acts, advs, rets = gather("obs", k), gather("acts", k), gather("advs", k), gather("rets", k)
                v_old, old_lp = gather("old_values", k), gather("old_logp", k)
                mean, tape_pf = ops.mlp_forward(self.pf_layers, obs, self.act)
                v, tape_vf = ops.mlp_forward(self.vf_layers, obs, self.act)
                d_mean, d_v = _C.ppo_generic_losses(
                    mean, algo.pf.logstd.detach(), acts, advs.view(-1), None if old_lp is None else old_lp.view(-1),
                    v.view(-1), rets.view(-1), None if v_old is None else v_old.view(-1), raw[k], n_global, *hyper,
                    self.g_logstd, info[k])
                ops.mlp_backward(tape_pf, d_mean, grads=self.gviews[0], workspace=ws)
                ops.mlp_backward(tape_vf, d_v, grads=self.gviews[1], workspace=ws)
                dist.all_reduce_sum_(self.grads)                       # C1: gradient SUM over ranks
                a = _C.AdamArgs()
                a.params, a.grads, a.exp_avg, a.exp_avg_sq = (self.flat.data_ptr(), self.grads.data_ptr(),
                                                              self.m.data_ptr(), self.v.data_ptr())
                a.n_groups = 2
                a.group_sizes[0], a.group_sizes[1] = self.P_pf, self.P_vf
                a.max_norm, a.beta1, a.beta2, a.eps, a.grad_scale = 0.5, 0.9, 0.999, 1e-5, 1.0
                a.step_count, a.norms_out = 0, norms[k].data_ptr()
                a.step_state, a.device_lr = self.step_state.data_ptr(), self.lr_dev.data_ptr()   # device-side step count / lr
                _C.clip_adam(a, dev)

        # eager on the first visit of a configuration, captured into a HIP graph on the second, replayed afterwards
        key = (K, rows_mb, N, rows_total, n_global, idx_dev.data_ptr(), stats.data_ptr(), pre_key, pre is not None) + hyper + tuple(
            0 if t.get(k_) is None else t[k_].data_ptr() for k_ in ("obs", "acts", "advs", "rets", "old_values", "old_logp"))
        graphs = self._graphs
        if not replayable or (key not in graphs and len(graphs) >= 4):
            launch_all()
        elif key in graphs:
            graphs[key].replay()
        elif key not in self._seen:
            self._seen.add(key)
            launch_all()
        else:
            graph, _ = _C.capture_graph(launch_all)
            graphs[key] = graph
            graph.replay()
        self.step_count += K
        for s in self._opt_steps:
            s.fill_(float(self.step_count))
        dist.reduce_info_(info)
        host = stats.cpu()                                             # the only host sync of the update
        make = self._infos_a2c if loss_mode == _C.LOSS_A2C else self._infos
        return make(host[:4 * K].view(K, 4).numpy(), host[4 * K:28 * K].view(K, 24).numpy(),
                    host[28 * K:].view(torch.float32).view(K, 2).numpy(), n_global)
