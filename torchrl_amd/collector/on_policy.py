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
    step) -- statistically equivalent, no host work.
  * `prefetch_noise=True` (host mode): the NEXT rollout's (T, N, A) block is drawn by
    worker threads into page-locked memory while the current iteration runs on the
    device, and moved to the device by extra workgroups of the rollout launch that runs meanwhile (no copy
    command, see _NoisePrefetcher) -- the same values in the same order from
    the same generator (nothing else on this path draws from it between two
    rollouts, torchrl/algo/on_policy/ppo.py:27-152), just earlier; what bench.py's
    headline uses.  See _NoisePrefetcher.
"""
import copy
import os

import numpy as np
import torch

from .. import _C
from .. import dist
from . import noise
from .base import VecCollector, BaseCollector, _EpochResult


class _NoisePrefetcher:
    """The reference's exploration-noise stream (CPU torch generator, distribution.py:60-76) one rollout ahead.

    `take(T, N, A)` returns the device tensor of this rollout's draws and immediately starts the draw of the NEXT block of
    the same shape: a worker thread fills a page-locked buffer with `torch.randn(out=...)` (the op releases the
    interpreter lock, and the block is cut into segments drawn by several threads at once: ~1 ms of host time for
    128 x 2048 x 6 instead of ~3, which would otherwise sit between two iterations); the NEXT `take` moves it to one of two
    device buffers (the worker threads never call into the HIP runtime) -- by a staging KERNEL on a stream of its own
    that reads the page-locked block in place and stamps a flag in device memory when it has landed; the rollout kernel
    polls that flag (`trl_rollout_t.noise_flag`), so the two streams share no event: the 0.15 ms transfer runs under the
    update that precedes the rollout.  (`staged = False`: a copy command on the rollout's own stream, in front of it.)  The
    draw depends on nothing the GPU produces, so the stream of values is the un-prefetched one, bit for bit.

    Guard: the worker draws from PRIVATE generators started at the state `_start` snapshotted on the main thread
    (collector/noise.py::draw_block) -- the default generator is read and set on the main thread only, in `take`.  A
    prefetched block is accepted when the default generator is still in the snapshotted state; the generator is then set
    to the end-of-block state, exactly where the in-place draws would leave it.  If the state differs -- someone seeded
    the generator or drew from it in between -- the block is dropped and drawn in place from the current state (for a
    re-seed that IS the reference order).  The default generator is never advanced speculatively, so `close()` or a
    shape change only has to wait for the worker: an abandoned prefetch leaves no trace in the stream.

    Env shards on several ranks: `layout = (n_total, e0)` -- the block is this rank's rows [e0, e0 + N) of every step's
    (n_total, A) draw, T chunks of the stream (noise.py), and the end-of-block state is the one after ALL ranks' rows."""

    def __init__(self, device):
        import threading
        self.device = torch.device(device)
        self._lock = threading.Lock()
        self._job = None                                   # dict(thread, shape, slot, state0, state1, event, error)
        self._host, self._dev = {}, {}
        self._free = {}                                    # per slot: the event of its last upload (page-locked buffer reusable)
        self._consumed = {}                                # per slot: the event behind the rollout that read the device buffer
        self._state, self._stamp = {}, {}                  # per slot: {stamp, arrival counter} on the device / last stamp
        self._side = None
        self.staged = True
        # carry = True: the block is moved to the device by a few extra workgroups of the PREVIOUS rollout launch
        # (trl_rollout_t.stage_*): per slot a page-locked {ready, ack} pair and a device {stamp, counter} pair
        self.carry = True
        self.wait_for_draw = False                         # test aid: publish the next block BEFORE the carrying launch goes out
        self.draw_threads = None                           # host threads per block (None: noise.default_threads())
        self.dropped_blocks = 0                            # prefetched blocks discarded because the generator had moved
        self.transport_counts = {"carried": 0, "staged": 0, "stream": 0}
        self._ctl, self._stg_state = {}, {}
        self._carrier = None                               # dict(id, key, event): the launch that carries the pending block
        self._ids = 0
        self._slot = 0

    def _buffers(self, shape, slot):
        key = (shape, slot)
        if key not in self._dev:
            n = int(np.prod(shape))
            self._host[key] = torch.empty(n).pin_memory()
            self._dev[key] = torch.empty(shape, device=self.device)
        return self._host[key], self._dev[key]

    def _draw_into(self, shape, slot, layout, state0):
        """Draw into the page-locked buffer of `slot` from generator state `state0`; returns the end-of-block state.  Host
        work only, on private generators: the worker thread never talks to the HIP runtime or to the default generator."""
        host, _ = self._buffers(shape, slot)
        T, n, a_dim = shape
        n_total, e0 = layout
        # one (T * N, A) draw == T successive (N, A) draws when N * A is a multiple of 16 (see _host_noise); the block itself
        # is produced by several host threads, each starting from the engine state at its segment (collector/noise.py)
        if n_total == n:
            return noise.draw_block(state0, host, threads=self.draw_threads)
        return noise.draw_block(state0, host, n_chunks=T, stride=n_total * a_dim, offset=e0 * a_dim, threads=self.draw_threads)

    def _upload(self, shape, slot, stream):
        """Page-locked block -> device.  Returns (device tensor, gate): gate = (flag tensor, stamp) the rollout has to
        wait for, or None when the copy sits on the rollout's own stream in front of it.
        Staged (default): `trl_stage_h2d_f32` on a side stream, no event between the streams -- an event dependency there
        costs a 5-50 ms host stall once every ~100 iterations on this runtime (measured in round 3,
        profiles/r03_grad_kernel_experiments.txt), which is why the first version of the prefetch kept the copy on the
        rollout's stream and paid its 0.13 ms per iteration."""
        host, dev = self._buffers(shape, slot)
        key = (shape, slot)
        if not self.staged:
            dev.copy_(host.view(shape), non_blocking=True)
            ev = self._free.get(key) or torch.cuda.Event()
            ev.record(stream)
            self._free[key] = ev
            return dev, None
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
        used = self._consumed.get(key)                      # the rollout that read this device buffer two blocks ago
        if used is not None and not used.query():
            used.synchronize()
        if key not in self._state:
            self._state[key] = torch.zeros(2, dtype=torch.int32, device=self.device)
            torch.cuda.current_stream(self.device).synchronize()          # (zeroed before the side stream's first launch)
        stamp = self._stamp[key] = (self._stamp.get(key, 0) % 0x7FFFFFF0) + 1
        _C.stage_h2d(host, dev.view(-1), self._state[key], stamp, self._side)
        ev = self._free.get(key) or torch.cuda.Event()
        ev.record(self._side)                               # the page-locked buffer is reusable behind the staging launch
        self._free[key] = ev
        return dev, (self._state[key], stamp)

    def consumed(self, shape, slot, stream):
        """Called behind the rollout that read the block: its device buffer may be refilled once that launch has run."""
        key = (tuple(int(v) for v in shape), slot)
        ev = self._consumed.get(key) or torch.cuda.Event()
        ev.record(stream)
        self._consumed[key] = ev

    def _worker(self):
        """One long-lived thread (no thread start per rollout): takes a job, draws, signals."""
        while True:
            job = self._queue.get()
            if job is None:
                return
            try:
                job["state1"] = self._draw_into(job["shape"], job["slot"], job["layout"], job["state0"])
                job["out"] = job["slot"]
                ctl = self._ctl.get((job["shape"], job["slot"]))
                if ctl is not None:
                    ctl[0] = job["id"]                      # the block is complete: the stagers of the carrying launch may go
            except BaseException as exc:                    # noqa: BLE001 -- reported by the consumer
                job["error"] = exc
            job["done"].set()

    def _start(self, shape, layout):
        import queue
        import threading
        if getattr(self, "_thread", None) is None:
            self._queue = queue.SimpleQueue()
            self._thread = threading.Thread(target=self._worker, name="trl-noise-prefetch", daemon=True)
            self._thread.start()
        slot = self._slot
        self._slot ^= 1
        self._buffers(shape, slot)                          # (allocated by this thread)
        free = self._free.get((shape, slot))
        if free is not None and not free.query():           # the upload that read this page-locked buffer two blocks ago
            free.synchronize()                              # (long finished; checked here so that the worker never waits on HIP)
        self._ids = self._ids % 0x7FFFFFF0 + 1
        if self.staged and self.carry and (shape, slot) not in self._ctl:
            self._ctl[(shape, slot)] = torch.zeros(2, dtype=torch.int32).pin_memory()
            self._stg_state[(shape, slot)] = torch.zeros(2, dtype=torch.int32, device=self.device)
        job = {"shape": shape, "layout": layout, "slot": slot, "state0": torch.get_rng_state(), "state1": None, "out": None,
               "error": None, "done": threading.Event(), "id": self._ids}
        self._job = job
        self._queue.put(job)

    def _drop(self):
        job, self._job = self._job, None
        if job is not None:
            job["done"].wait()                              # (the default generator never moved: nothing to give back)

    def _carried_ok(self, job):
        """Did the launch that carried `job`'s block stage it?  (Its stagers acknowledge in page-locked memory.)"""
        c = self._carrier
        if c is None or c["id"] != job["id"]:
            return False
        ctl = self._ctl[c["key"]]
        if int(ctl[1]) != job["id"]:
            c["event"].synchronize()                        # the carrying launch has run (normally long ago)
            if int(ctl[1]) != job["id"]:                    # its stagers gave up (the block was published too late):
                self._stg_state[c["key"]].zero_()           # their arrival counter may be half way
                return False
        return True

    def _settle_carrier(self):
        """No stager of an abandoned block may still be writing when the buffer is refilled another way."""
        c, self._carrier = self._carrier, None
        if c is not None:
            c["event"].synchronize()

    def carried(self, request, stream):
        """Called behind the rollout launch that was handed `request` (see `take`)."""
        ev = torch.cuda.Event()
        ev.record(stream)
        self._carrier = {"id": request["id"], "key": request["key"], "event": ev}
        self._free[request["key"]] = ev                     # (that launch reads the page-locked block)

    def take(self, n_steps, n, a_dim, stream, layout=None):
        """(device block, slot, gate, request): gate = (flag tensor, stamp) the rollout must wait for (None: the block is
        stream-ordered in front of it); request (or None) = what the rollout launch shall stage for the NEXT call.
        layout = (n_total, e0): this rank's env block within the per-step draw for all envs (default: all of it)."""
        shape = (int(n_steps), int(n), int(a_dim))
        layout = (int(n), 0) if layout is None else (int(layout[0]), int(layout[1]))
        job = self._job
        out = None
        cur = torch.get_rng_state()
        if job is not None:
            job["done"].wait()
            self._job = None
            if job["error"] is not None:
                raise job["error"]
            if job["shape"] == shape and job["layout"] == layout and torch.equal(cur, job["state0"]):
                out = used = job["slot"]
                torch.set_rng_state(job["state1"])         # where the in-place draws of this block leave the generator
            else:
                self.dropped_blocks += 1                   # the generator moved (or the shape changed): draw in place below
                if self.dropped_blocks == 1:
                    import logging
                    logging.getLogger("torchrl_amd").info(
                        "exploration-noise prefetch: the CPU generator was used or re-seeded between two rollouts (or the "
                        "rollout shape changed); that block is drawn in place, in the reference's order")
        if out is not None and self._carried_ok(job):
            dev, gate = self._buffers(shape, used)[1], (self._stg_state[(shape, used)], job["id"])
            self._carrier = None
            self.transport_counts["carried"] += 1
        else:
            self._settle_carrier()
            if out is None:                                 # first call / generator touched in between: draw in place
                used = self._slot
                self._slot ^= 1
                free = self._free.get((shape, used))
                if free is not None:
                    free.synchronize()
                torch.set_rng_state(self._draw_into(shape, used, layout, cur))
            dev, gate = self._upload(shape, used, stream)
            self.transport_counts["staged" if gate is not None else "stream"] += 1
        self._start(shape, layout)                          # the next block, under this iteration's device work
        request = None
        nxt = self._job
        if self.wait_for_draw and nxt is not None:
            nxt["done"].wait()
        if self.staged and self.carry and nxt is not None:
            key = (shape, nxt["slot"])
            host, ndev = self._buffers(shape, nxt["slot"])
            request = {"id": nxt["id"], "key": key, "src": host, "dst": ndev, "ctl": self._ctl[key],
                       "state": self._stg_state[key]}
        return dev, used, gate, request

    def close(self):
        self._drop()


class VecOnPolicyCollector(VecCollector):
    def __init__(self, vf, discount=0.99, noise_mode="host", prefetch_noise=False, **kwargs):
        self.vf = vf
        super().__init__(noise_mode=noise_mode, **kwargs)
        self.discount = discount
        self.prefetch_noise = bool(prefetch_noise) or os.environ.get("TRL_PREFETCH_NOISE") == "1"
        self._prefetcher = None
        self._check_shapes()

    def stop_noise_prefetch(self):
        """Drop a speculatively drawn block (the CPU generator was never advanced for it: it already is where the
        un-prefetched path would be)."""
        if self._prefetcher is not None:
            self._prefetcher.close()

    def terminate(self):
        self.stop_noise_prefetch()
        super().terminate()

    @property
    def funcs(self):
        return {"pf": self.pf, "vf": self.vf}

    def _check_shapes(self):
        """`_spec` = (D, H, A, act) when the persistent rollout kernel is instantiated for these networks; None
        otherwise -- any other MLP shape is collected by the per-step launch sequence on the generic dense-layer
        kernels."""
        from .. import ops
        if not hasattr(self.pf, "logstd"):
            raise _C.TrlError("the on-policy collector supports GuassianContPolicyBasicBias policies")
        ps = self.pf.mlp2_spec() if hasattr(self.pf, "mlp2_spec") else None
        vs = self.vf.mlp2_spec() if hasattr(self.vf, "mlp2_spec") else None
        self._act = ops.act_code(self.pf)
        self._dims = (int(ops.linear_layers(self.pf)[0][0].shape[1]), int(self.pf.logstd.numel()))
        if self._dims != (self.env.obs_dim, self.env.act_dim) or int(ops.linear_layers(self.vf)[0][0].shape[1]) != self.env.obs_dim:
            raise _C.TrlError("policy / value input and output sizes %s do not match the env (%d obs, %d act)"
                              % (self._dims, self.env.obs_dim, self.env.act_dim))
        # the fused 2-layer forward (and the cooperative, normalised rollout) are instantiated for the benchmark shape
        # (trl_mlp2_forward_supported); the persistent rollout kernel itself also carries every other 64-wide two-layer
        # pair with 2..32 inputs and 1..8 actions through its runtime-dims instantiations (trl_rollout_supported) -- the
        # shapes the fused UPDATE kernels carry (trl_ppo_partial_stride > 0); anything else is collected by the per-step
        # launch sequence on the dense-layer kernels
        lib = _C.lib()
        pair = ps is not None and vs is not None and vs[:2] == ps[:2] and vs[2] == 1 and vs[3] == ps[3] and \
            os.environ.get("TRL_GENERIC_PPO") != "1"
        mlp2 = pair and lib.trl_mlp2_forward_supported(ps[0], ps[1], ps[2]) and lib.trl_mlp2_forward_supported(ps[0], ps[1], 1)
        self._mlp2 = ps if mlp2 else None                                   # fused 2-layer forward kernel usable
        # (TRL_NO_RT_ROLLOUT=1 switches the RUNTIME-dims instantiations off for A/B measurements; the compile-time benchmark
        # shape -- the one the fused forward is instantiated for -- keeps its rollout kernel)
        roll = pair and bool(lib.trl_rollout_supported(ps[0], ps[1], ps[2], ps[3])) and \
            (bool(mlp2) or os.environ.get("TRL_NO_RT_ROLLOUT") != "1")
        self._spec = ps if (roll and not getattr(self.env, "is_host_env", False)) else None   # ... and the rollout kernel

    def _forward(self, net, x, out_dim, out=None):
        """mean / value of an MLP on the device: the fused 2-layer kernel when instantiated, the dense-layer family
        otherwise."""
        from ..networks import nets as _nets
        _nets.settle(net, x.device)                                     # (parameters still being stepped on another stream)
        if self._mlp2 is not None:
            D, H, A, act = self._mlp2
            return _C.mlp2_forward(net.flat_params(), x, D, H, out_dim, act, out=out)
        from .. import ops
        y, _ = ops.mlp_forward(ops.linear_layers(net), x, self._act)
        if out is not None:
            out.copy_(y)
            return out
        return y

    # ---- launch ----
    def _fused_norm_ok(self, env, update):
        """The persistent kernel carries a normalised env when its workgroups can all be resident (they meet once
        per step to pool the observation statistics) and the statistics are not shared between GPUs."""
        if not hasattr(env, "_obs_normalizer") or getattr(self, "force_per_step", False) or getattr(env, "is_host_env", False):
            return False
        if self._mlp2 is None:                                              # (the cooperative kernel: benchmark shape only)
            return False
        if update and dist.collectives_active():
            return False
        if getattr(self, "_norm_cap", None) is None:
            D, H, A, act = self._spec
            self._norm_cap = _C.lib().trl_rollout_norm_max_envs(D, H, A, act)
        return (not update) or env.env_nums <= self._norm_cap

    def _launch(self, env, n_steps, store, deterministic, noise, max_frames=None, policy_ob=None, publish=False,
                noise_gate=None, stage=None):
        D, H, A, act = self._spec
        buf = self.replay_buffer
        a = _C.RolloutArgs()
        nz = getattr(env, "_obs_normalizer", None)
        if nz is not None:
            N_ = env.env_nums
            if getattr(self, "_norm_ws", None) is None or self._norm_ws_n != N_:
                self._norm_ws = torch.zeros(_C.lib().trl_rollout_norm_workspace(N_), dtype=torch.float64, device=env.device)
                self._norm_ws_n = N_
            update = bool(env.training and nz.should_estimate)
            a.norm_state, a.policy_obs = nz.state.data_ptr(), policy_ob.data_ptr()
            a.norm_workspace, a.norm_clip = self._norm_ws.data_ptr(), float(nz.clip)
            a.norm_update = int(update)
            a.normalize_partial_reset = int(bool(getattr(env, "normalize_partial_reset", False)))
        a.pf_params = self.pf.flat_params().data_ptr()
        a.vf_params = self.vf.flat_params().data_ptr()
        # The rollout reads the policy only; the value function may still be in the hands of the previous epoch's critic
        # updates on another stream (algo/on_policy/ppo.py: two update chains): the value pass behind the rollout waits
        # for their end, the rollout itself does not.
        from ..networks import nets as _nets
        _nets.settle(self.pf, env.device)
        vf_ev = _nets.pending_event(self.vf)
        if vf_ev is not None and store:
            # ... and that chain still READS the stored observations: a rollout that rewrites the whole ring gets the
            # shadow `obs` tensor (swapped in here, before the pointers are taken) and runs beside it; one that rewrites
            # only part of the ring has to keep the other rows, so it waits for the chain like everybody else.
            if n_steps == buf._max_replay_buffer_size and hasattr(buf, "_obs"):
                buf._flip_key("obs")
                a.value_wait_event = vf_ev.cuda_event
                self._vf_event_keepalive = vf_ev                        # (the handle must outlive the launch)
            else:
                _nets.settle(self.vf, env.device)
        a.D, a.H, a.A, a.act = D, H, A, act
        a.tanh_action = int(bool(self.pf.tanh_action))
        a.env_A, a.env_B = env.env_A.data_ptr(), env.env_B.data_ptr()
        a.reward_scale, a.horizon, a.env_seed_base = env.effective_reward_scale, env.horizon, env.seed_base
        a.cur_obs, a.t_env, a.cur_step = env.cur_obs.data_ptr(), env.t_env.data_ptr(), env.cur_step.data_ptr()
        a.episode_idx, a.ep_return = env.episode_idx.data_ptr(), env.ep_return.data_ptr()
        a.noise = noise.data_ptr() if noise is not None else None
        if noise is not None and noise_gate is not None:               # staged on another stream: the kernel waits for the stamp
            a.noise_flag, a.noise_stamp = noise_gate[0].data_ptr(), int(noise_gate[1])
        if stage is not None:                                          # the NEXT rollout's block rides in on this launch
            a.stage_src, a.stage_dst, a.stage_n = stage["src"].data_ptr(), stage["dst"].data_ptr(), stage["src"].numel()
            a.stage_ready, a.stage_ack = stage["ctl"].data_ptr(), stage["ctl"].data_ptr() + 4
            a.stage_job, a.stage_state = int(stage["id"]), stage["state"].data_ptr()
        a.noise_step0 = self.global_step
        a.deterministic = int(deterministic)
        N = env.env_nums
        if store:
            feats = (("obs", D), ("next_obs", D), ("acts", A), ("values", 1), ("rewards", 1),
                     ("terminals", 1), ("time_limits", 1), ("old_logp", 1))
            for key, f in feats:
                setattr(a, key, buf._ensure_key(key, (N, f)).data_ptr())
            a.rows, a.top = buf._max_replay_buffer_size, buf._top
            # a rollout that ends in the ring's last row also leaves V(next_obs) of that row: the epoch's bootstrap value
            boot = (buf._top + n_steps) % buf._max_replay_buffer_size == 0 and n_steps > 0
            if boot:
                if getattr(buf, "_boot", None) is None or buf._boot.shape[0] != N:
                    buf._boot = torch.empty(N, 1, device=env.device)
                a.boot_values = buf._boot.data_ptr()
        else:
            a.rows, a.top = 1, 0
        a.N, a.n_steps = N, n_steps
        a.max_episode_frames = int(self.max_episode_frames if max_frames is None else max_frames)
        a.discount = float(self.discount)
        a.epoch_reward, a.ep_count, a.ep_log = (self._epoch_reward.data_ptr(), self._ep_count.data_ptr(),
                                                self._ep_log.data_ptr())
        a.ep_cap, a.step0 = self.EP_LOG_CAP, 0
        nxt = self._clear_header(swap=True)                            # (may switch headers: read the pointers after it)
        a.epoch_reward, a.ep_count = self._epoch_reward.data_ptr(), self._ep_count.data_ptr()
        a.clear_header = nxt.data_ptr()
        # both headers + the speculative head of the episode log are written to their page-locked twin by the value pass
        # (train_one_epoch then needs no copy command behind the launch)
        self._published = False
        if store and publish and self._blob_host is not None:
            words = 8 + 3 * self.SPECULATIVE_ROWS
            a.publish_dst, a.publish_src, a.publish_words = self._blob_host.data_ptr(), self._blob.data_ptr(), words
            self._published = True
        _C.rollout(a, env.device)
        self._idle_hdr_clean = True
        if store:
            buf._advance(n_steps)
            buf._boot_fresh = bool(boot)
            # log pi_old written by the kernel covers the whole ring only for a full-ring launch
            buf._old_logp_fresh = (n_steps == buf._max_replay_buffer_size)

    def _noise_layout(self, env):
        """(n_total, e0): the env block [e0, e0 + N) this rank owns of the n_total envs whose (n_total, A) noise tensor the
        reference draws per step (distribution.py:60-76)."""
        n, w = env.env_nums, dist.world_size()
        if w == 1:
            return n, 0
        n_total, e0 = int(getattr(env, "total_env_nums", n * w)), int(getattr(env, "index_offset", dist.rank() * n))
        if n_total == n:
            # envs built without total_env_nums / index_offset on a multi-rank run (the env's default is its own count): every
            # rank would take the one-process path and draw the SAME block.  Same convention as the device-noise and
            # epsilon-greedy paths (dist.shard_rows_of_global): rank r owns rows [r * n, (r + 1) * n) of an n * w draw.
            n_total, e0 = n * w, dist.rank() * n
        return n_total, e0

    def _host_noise(self, n_steps, env):
        """The reference's CPU stream, step by step; with env shards on several ranks, this rank's rows of each draw."""
        A = self._dims[1]
        n = env.env_nums
        n_total, e0 = self._noise_layout(env)
        if n_total == n and (n * A) % 16 == 0 and noise.fast_path_ok():
            # torch's CPU normal_ draws its uniforms sequentially over the whole tensor and transforms them in blocks of
            # 16, so ONE (n_steps * N, A) draw is bit-identical to n_steps successive (N, A) draws whenever N * A is a
            # multiple of 16 (noise.fast_path_ok() checks it on this torch build) -- a third of the
            # host time of the per-step loop; the block is cut into segments that several host threads draw at once, each from
            # the engine state at its first element (collector/noise.py: same values, same generator state afterwards)
            block = noise.randn_into(torch.empty(n_steps * n, A))
            return block.view(n_steps, n, A).to(env.device, non_blocking=True).contiguous()
        if n_total != n and noise.shard_ok(n, n_total, e0, A):
            # env shards: only this rank's rows of every step's (n_total, A) draw are produced -- from the engine state at
            # their position in the stream -- and the generator ends where the draws for ALL envs would leave it
            block = noise.randn_shard_into(torch.empty(n_steps, n, A), n_steps, n, n_total, e0, A)
            return block.to(env.device, non_blocking=True).contiguous()
        make = lambda m, f: torch.randn(m, f)
        draws = [dist.shard_rows_of_global(make, 1, n, A, "cpu").cpu() for _ in range(n_steps)]
        return torch.stack(draws).to(env.device, non_blocking=True).contiguous()

    def _can_prefetch(self, n_steps):
        """Prefetch covers what the derived-state draws cover: this rank's block of every step's draw aligned to the
        generator's groups of 16 (see _host_noise), and whole rollouts (a one-step take_actions keeps the per-step draw)."""
        if not self.prefetch_noise or n_steps <= 1:
            return False
        n, A = self.env.env_nums, self._dims[1]
        n_total, e0 = self._noise_layout(self.env)
        if n_total == n:
            return (n * A) % 16 == 0 and noise.fast_path_ok()
        return noise.shard_ok(n, n_total, e0, A)

    # ---- per-step launch sequence: envs with a running observation normaliser ----
    def _step_buffers(self, env):
        sb = getattr(self, "_sb", None)
        if sb is None or sb["N"] != env.env_nums:
            D, A = self._dims
            N, dev = env.env_nums, env.device
            f = lambda *shape: torch.empty(*shape, device=dev)
            sb = self._sb = {"N": N, "mean": f(N, A), "eps": f(N, A), "nxt_raw": f(N, D), "v_next": f(N, 1),
                             "done": f(N, 1), "any": torch.zeros(1, dtype=torch.int32, device=dev),
                             # rows used when nothing is stored (evaluation)
                             "obs": f(N, D), "next_obs": f(N, D), "acts": f(N, A), "values": f(N, 1), "rewards": f(N, 1),
                             "terminals": f(N, 1), "time_limits": f(N, 1), "old_logp": f(N, 1)}
        return sb

    def _step_launches(self, env, ob, store, deterministic, noise_t, step, max_frames=None):
        """One take_actions (torchrl/collector/on_policy.py:90-155) as ~12 launches, for envs the persistent rollout
        kernel cannot carry: a running observation normaliser shared between GPUs or too large for one co-resident
        grid, and host Python envs (`torchrl_amd.env.VecEnv`).  `ob` is what the policy sees (normalised, or raw
        right after a reset -- the reference's Q14); returns the next policy input."""
        D, A = self._dims
        N, buf, sb = env.env_nums, self.replay_buffer, self._step_buffers(env)
        nz = getattr(env, "_obs_normalizer", None)
        if store:
            row = buf._top
            feats = (("obs", D), ("next_obs", D), ("acts", A), ("values", 1), ("rewards", 1), ("terminals", 1),
                     ("time_limits", 1), ("old_logp", 1))
            r = {k: buf._ensure_key(k, (N, w))[row] for k, w in feats}
        else:
            r = sb
        r["obs"].copy_(ob)
        mean = self._forward(self.pf, ob, A, out=sb["mean"])
        self._forward(self.vf, ob, 1, out=r["values"])
        if deterministic:
            eps = None
        elif noise_t is not None:
            eps = noise_t
        elif dist.world_size() == 1:
            eps = _C.philox_normal(sb["eps"], self._noise_seed, self.global_step)
        else:                                                             # this rank's rows of the draw for ALL envs
            make = lambda m, f: _C.philox_normal(torch.empty(m, f, device=env.device), self._noise_seed, self.global_step)
            eps = dist.shard_rows_of_global(make, 1, N, A, env.device)
        _C.gauss_explore(mean, self.pf.logstd.detach(), eps, bool(self.pf.tanh_action), act=r["acts"],
                         logp=r["old_logp"].view(N))
        raw_next = sb["nxt_raw"] if nz is not None else r["next_obs"]
        self._env_advance(env, r["acts"], raw_next, r["rewards"], sb["done"], r["time_limits"])
        if nz is not None:
            nz.update_filt(raw_next, update=env.training, out=r["next_obs"])            # NormObs.observation
        self._forward(self.vf, r["next_obs"], 1, out=sb["v_next"])
        sb["any"].zero_()
        _C.onpolicy_bookkeep(r["rewards"], sb["done"], sb["v_next"], self.discount, r["terminals"], env.cur_step,
                             env.ep_return, self.max_episode_frames if max_frames is None else max_frames, self._mask,
                             sb["any"], self._epoch_reward, self._ep_count, self._ep_log, step)
        self._env_reset_masked(env, r["next_obs"] if (store and nz is None) else None)
        # partial_reset returns the RAW observations of all envs (base_wrapper.py:23-26, vecenv.py:47-51)
        alt = nz.filt(env.cur_obs) if (nz is not None and getattr(env, "normalize_partial_reset", False)) else env.cur_obs
        nxt = torch.empty(N, D, device=env.device)
        _C.select_on_flag(sb["any"], alt, r["next_obs"], nxt)
        if store:
            buf._advance()
        return nxt

    def _rollout_per_step(self, n_steps):
        """`n_steps` vector steps as per-step launch sequences.  On a device env with device noise and a ring that the
        rollout fills exactly, the whole sequence (~17 launches x n_steps) is captured into a HIP graph on its second
        visit and replayed afterwards -- the eager sequence is host-launch-bound (~190 us per step); the rollout's
        exploration noise is then drawn up front in one Philox launch."""
        env, buf = self.env, self.replay_buffer
        D, A = self._dims
        graphable = (self.noise_mode == "device" and not getattr(env, "is_host_env", False)
                     and buf._top == 0 and n_steps == buf._max_replay_buffer_size)
        # same launches and the same noise whether captured or not; env shards on several ranks (collectives between
        # the steps) run it eagerly on their block of the global draw
        capture = os.environ.get("TRL_NO_GRAPH") != "1" and dist.world_size() == 1
        ob = torch.as_tensor(self.current_ob).to(device=env.device, dtype=torch.float32).contiguous()
        if not graphable:
            self._clear_header()
            noise = self._host_noise(n_steps, env) if self.noise_mode == "host" else None
            for t in range(n_steps):
                ob = self._step_launches(env, ob, True, False, None if noise is None else noise[t], t)
                self.global_step += 1
        else:
            st = getattr(self, "_roll_graph", None)
            if st is None or st["key"] != (n_steps, env.env_nums):
                st = self._roll_graph = {"key": (n_steps, env.env_nums), "graph": None, "seen": False, "hdr": self._hdr_i,
                                         "ob0": torch.empty(env.env_nums, D, device=env.device),
                                         "noise": torch.empty(n_steps, env.env_nums, A, device=env.device), "out": None}
            self._clear_header(index=st["hdr"])                           # (a captured sequence carries its header's address)
            st["ob0"].copy_(ob)
            if dist.world_size() == 1:
                _C.philox_normal(st["noise"], self._noise_seed, self.global_step)
            else:                                                         # (T, N_total, A) drawn identically on every rank
                make = lambda m, f: _C.philox_normal(torch.empty(m, f, device=env.device), self._noise_seed, self.global_step)
                st["noise"].copy_(dist.shard_rows_of_global(make, n_steps, env.env_nums, A, env.device).view_as(st["noise"]))

            def steps():
                o = st["ob0"]
                for t in range(n_steps):
                    o = self._step_launches(env, o, True, False, st["noise"][t], t)
                buf._top = 0                                              # (the host-side ring cursor is advanced below)
                return o
            if st["graph"] is not None:
                st["graph"].replay()
            elif not st["seen"] or not capture:
                st["seen"] = True
                st["out"] = steps()
            else:
                st["graph"], st["out"] = _C.capture_graph(steps)
                graph = st["graph"]
                graph.replay()
            ob = st["out"]
            self.global_step += n_steps
            buf._advance(n_steps)
        self.current_ob = ob
        buf._old_logp_fresh = (n_steps == buf._max_replay_buffer_size)

    def rollout(self, n_steps):
        """Enqueue `n_steps` vector steps into the replay buffer; no host sync."""
        self._resolve_pending()                                            # the header / episode log are about to be reused
        self.env.train()
        if getattr(self.env, "is_host_env", False) or self._spec is None:  # host Python envs / shapes without a fused kernel:
            return self._rollout_per_step(n_steps)                           # per-step launch sequence
        if hasattr(self.env, "_obs_normalizer"):
            nz = self.env._obs_normalizer
            if not self._fused_norm_ok(self.env, self.env.training and nz.should_estimate):
                return self._rollout_per_step(n_steps)
            ob = torch.as_tensor(self.current_ob).to(device=self.env.device, dtype=torch.float32).contiguous().clone()
            noise = self._host_noise(n_steps, self.env) if self.noise_mode == "host" else None
            self._launch(self.env, n_steps, True, False, noise, policy_ob=ob, publish=True)
            self.global_step += n_steps
            self.current_ob = ob
            self._check_rendezvous = True                       # read with the epoch header (no sync here)
            return
        slot = None
        if self.noise_mode == "host" and self._can_prefetch(n_steps):
            if self._prefetcher is None:
                self._prefetcher = _NoisePrefetcher(self.env.device)
            stream = torch.cuda.current_stream(self.env.device)
            noise, slot, gate, request = self._prefetcher.take(n_steps, self.env.env_nums, self._dims[1], stream,
                                                               layout=self._noise_layout(self.env))
        else:
            noise, gate, request = (self._host_noise(n_steps, self.env) if self.noise_mode == "host" else None), None, None
        self._noise_gated = gate is not None
        self._launch(self.env, n_steps, True, False, noise, publish=True, noise_gate=gate, stage=request)
        if slot is not None:
            stream = torch.cuda.current_stream(self.env.device)
            self._prefetcher.consumed(noise.shape, slot, stream)
            if request is not None:
                self._prefetcher.carried(request, stream)
        self.global_step += n_steps
        self.current_ob = self.env.cur_obs

    def _rendezvous_check(self):
        if getattr(self, "_noise_gated", False) and self.train_epoch_reward != self.train_epoch_reward:
            # the kernel poisons the epoch reward when its wait for the staged noise block times out (k_rollout.hip): that
            # rollout read a stale block -- an error, not data (the reference stops on NaN too, collector/on_policy.py:102-107)
            import logging
            msg = ("rollout: NaN epoch reward -- the staged exploration-noise block never arrived (stamp wait timed out) or "
                   "the policy produced NaN actions.  The collected rollout is not valid data; an update that was launched "
                   "on it before this result was read (RLAlgo.train launches the update first and reads the epoch result "
                   "after) HAS changed the parameters: restore the last snapshot before continuing")
            logging.getLogger("torchrl_amd").error(msg)
            raise _C.TrlError(msg)
        if getattr(self, "_check_rendezvous", False):
            self._check_rendezvous = False
            if int(self._norm_ws[:2].view(torch.int32)[2].item()) != 0:
                raise _C.TrlError("rollout: grid rendezvous timed out (workgroups were not co-resident)")

    def take_actions(self):
        self.rollout(1)
        return float(self._epoch_reward.item())

    def eval_one_epoch(self):
        """Greedy evaluation (torchrl/collector/base.py:232-280): every eval env plays its
        first episode with action = tanh(mean); nothing is written to the replay buffer."""
        self._resolve_pending()
        env = self.eval_env
        if hasattr(self.env, "_obs_normalizer"):                           # collector/base.py:236-237
            env._obs_normalizer = copy.deepcopy(self.env._obs_normalizer)
        env.eval()
        rews, lens = [], []
        for _ in range(self.eval_episodes):
            ob = env.reset()
            if getattr(env, "is_host_env", False) or self._spec is None:
                self._clear_header()
                for t in range(self._eval_steps(env)):
                    ob = self._step_launches(env, ob, False, True, None, t, max_frames=2 ** 31 - 1)
                    if len({int(i) for _, i, _ in self._finished_episodes()}) == env.env_nums:
                        break                                               # every env finished its first episode
            elif hasattr(env, "_obs_normalizer") and self._fused_norm_ok(env, False):
                self._launch(env, env.horizon, False, True, None, max_frames=2 ** 31 - 1, policy_ob=ob.contiguous().clone())
            elif hasattr(env, "_obs_normalizer"):
                self._clear_header()
                for t in range(env.horizon):
                    ob = self._step_launches(env, ob, False, True, None, t, max_frames=2 ** 31 - 1)
            else:
                self._launch(env, env.horizon, False, True, None, max_frames=2 ** 31 - 1)
            log = self._finished_episodes()
            first = {}
            for step, idx, ret in log:
                first.setdefault(int(idx), (ret, int(step) + 1))
            rews += [np.float32(first[i][0]) for i in sorted(first)]
            lens += [first[i][1] for i in sorted(first)]
        return {"eval_rewards": rews, "eval_traj_length": float(np.mean(lens)) if lens else 0.0}


class OnPolicyCollectorBase(VecOnPolicyCollector):
    """The reference's single-env on-policy collector (torchrl/collector/on_policy.py:8-60): here simply the vector
    collector on a one-env device env (`torchrl.env.get_env`), so `examples/ppo_continuous.py` runs on the same
    kernels (SURVEY.md 8(d) cfg 1)."""

    def __init__(self, vf, discount=0.99, **kwargs):
        super().__init__(vf, discount=discount, **kwargs)
        if self.env.env_nums != 1:
            raise _C.TrlError("OnPolicyCollectorBase drives a single env (torchrl.env.get_env); "
                              "use VecOnPolicyCollector for %d envs" % self.env.env_nums)
