#!/usr/bin/env python
"""Headline benchmark: env-steps/s of the full PPO iteration on the synthetic
HalfCheetah-shaped workload (BASELINE.json cfg 2).

One "step" = one PPO iteration = collect T=128 steps on N=2048 envs per GPU,
GAE, then opt_epochs=10 x 4 minibatch updates of B=65536 (per GPU) -- everything
the reference's RLAlgo.train loop body does between evaluations
(torchrl/algo/rl_algo.py:111-118), inputs resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --gpus N ...          (no launcher: the script spawns its N ranks itself, one per GPU)

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

N_PER_GPU, T, BATCH_PER_GPU, OPT_EPOCHS = 2048, 128, 65536, 10
D, H, A = 17, 64, 6
# algorithmic FLOPs per sample of the fused minibatch-gradient kernel (2*MACs):
# pf fwd 11136 + vf fwd 10496 + backward 2x(fwd) for both nets; log pi_old is cached
# by the collector, so target_pf is not re-run (76032 - 11136)
FLOP_PER_SAMPLE = 3 * (11136 + 10496)
MFMA_F32_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md


class CountingLogger:
    """Takes the per-update info dicts the way torchrl_amd.utils.Logger does between two rows -- `add_update_info(dict)`, or
    `add_update_infos_later(resolve)` for updates that were launched but not waited for -- and counts them; `drain()` reads
    whatever is still outstanding (`iteration` calls it once per iteration, the timed region once more before its clock
    stops: every iteration's statistics are read inside the timed region)."""

    def __init__(self):
        self.updates, self._later = 0, []

    def add_update_info(self, d):
        self.drain()
        self.updates += 1

    def add_update_infos_later(self, resolve):
        self._later.append(resolve)

    def drain(self, leave=0):
        """leave = 1: everything but the update launched last (whose statistics are still being produced)."""
        cut = max(len(self._later) - leave, 0)
        later, self._later = self._later[:cut], self._later[cut:]
        for resolve in later:
            self.updates += len(resolve())

    def add_epoch_info(self, *a, **k): pass
    def log(self, *a): pass
    def finish(self): pass


def build_agent(dev, world, rank, seed=0, noise_mode="device"):
    import torch
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.algo import PPO
    from torchrl.collector.on_policy import VecOnPolicyCollector
    from torchrl.env import get_vec_env
    from torchrl.replay_buffers.on_policy import OnPolicyReplayBuffer
    torch.manual_seed(seed)
    np.random.seed(seed)                         # identical index streams on every rank
    net = dict(hidden_shapes=[H, H], append_hidden_shapes=[], base_type=networks.MLPBase,
               activation_func=torch.nn.Tanh)
    pf = policies.GuassianContPolicyBasicBias(input_shape=D, output_shape=A, tanh_action=True, **net)
    vf = networks.Net(input_shape=(D,), output_shape=1, **net)
    kw = dict(device=dev, index_offset=rank * N_PER_GPU, total_env_nums=world * N_PER_GPU)
    env = get_vec_env("SynthHalfCheetah-v0", {"reward_scale": 1, "obs_norm": False}, N_PER_GPU, **kw)
    eval_env = get_vec_env("SynthHalfCheetah-v0", {"reward_scale": 1, "obs_norm": False}, N_PER_GPU, **kw)
    env.seed(seed)
    buf = OnPolicyReplayBuffer(N_PER_GPU * T, env_nums=N_PER_GPU, time_limit_filter=True, device=dev)
    col = VecOnPolicyCollector(vf, env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=dev,
                               epoch_frames=N_PER_GPU * T, max_episode_frames=1000, noise_mode=noise_mode)
    agent = PPO(pf=pf, vf=vf, plr=3e-4, vlr=3e-4, clip_para=0.2, opt_epochs=OPT_EPOCHS, tau=0.95, shuffle=True,
                entropy_coeff=0.005, discount=0.99, num_epochs=100000, batch_size=BATCH_PER_GPU, gae=True,
                env=env, replay_buffer=buf, collector=col, logger=CountingLogger(), device=dev, save_dir=None)
    return agent, col


def iteration(agent, col, epoch):
    """One pass of the reference's loop body (torchrl/algo/rl_algo.py:111-118): `collector.train_one_epoch()` --
    the rollout plus the per-epoch read-back of the epoch reward and the finished-episode list
    (collector/on_policy.py:277-286 here) -- then `update_per_epoch()` = GAE + opt_epochs x minibatch updates with one
    host read of the update statistics.  That read is pipelined by one iteration, as in RLAlgo.train with the package's
    Logger (the dicts are taken when the next log row needs them): the update is launched, and its 40 info dicts are read
    and assembled while the NEXT rollout runs on the device instead of while the device idles."""
    collected = col.train_one_epoch()
    agent.current_epoch = epoch
    agent.update_per_epoch()                      # launched; its statistics are read in the next iteration (or at the end)
    agent.logger.drain(leave=1)                   # the PREVIOUS update's statistics, assembled while this rollout runs
    return len(collected["train_rewards"]), collected["train_epoch_reward"]   # consumed where RLAlgo.train consumes it


def log(msg):
    print("[bench %7.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


FLOP_PER_ENV_STEP = 671e3                         # whole iteration, SURVEY.md section 8(d) (log pi_old cached)
BYTES_PER_ENV_STEP = 1240                         # algorithmic HBM bytes per env-step (collect 176 + GAE 24 + 10 x 104), BASELINE.md section 4
PROBE_STEPS = 5                                   # iterations of the event-probed pass after a graph-replayed timed region


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(budget_s=27.0):
    """The reference-style CPU path (oracle/, a port), timed on this host on a BOUNDED sample of the
    same workload and scaled to one full iteration: SubProcVecEnv-like stepping of per-env Python
    objects over pickled pipes + torch-CPU policy/value forward + numpy fp64 ring (collect leg, timed at the two
    worker counts SURVEY.md section 8(d) names: proc_nums = 4, the value the reference's example hard-codes
    (examples/ppo_continuous_vec_subproc.py:31-36), and proc_nums = the host's hardware threads, rounded down to a
    power of two that divides N; as many of the T=128 vector steps as fit in a third of the budget each), then
    torch-CPU PPO minibatch updates of the full B=65536 on synthetic rollout data (update leg: as many of the 40 as
    fit in the last third).  `value` is the faster of the two settings."""
    import functools
    import torch
    from oracle import nets, replay
    from oracle.collector import VecOnPolicyCollectorOracle
    from oracle.subproc_env import SubProcVecEnvCPU
    from oracle.ppo import PPOOracle
    from oracle.synth_env import SynthSingleEnvCPU
    cores = os.cpu_count() or 1
    proc_hi = 1
    while proc_hi * 2 <= cores and N_PER_GPU % (proc_hi * 2) == 0:
        proc_hi *= 2
    threads = min(cores, 32)
    torch.set_num_threads(threads)
    fns = [functools.partial(SynthSingleEnvCPU, i) for i in range(N_PER_GPU)]
    gen = torch.Generator().manual_seed(0)
    pf, vf = nets.init_mlp(D, [H, H], A, generator=gen), nets.init_mlp(D, [H, H], 1, generator=gen)
    ls = torch.full((A,), float(np.log(0.125)))
    collect = {}
    for procs in sorted({min(4, proc_hi), proc_hi}):
        log("cpu baseline: spawning %d env workers" % procs)
        t_spawn = time.perf_counter()
        env = SubProcVecEnvCPU(procs, N_PER_GPU, fns, SynthSingleEnvCPU(0))
        try:
            ring = replay.RingOracle(N_PER_GPU * T, env_nums=N_PER_GPU, time_limit_filter=True)
            col = VecOnPolicyCollectorOracle(env, ring, pf, ls, vf, epoch_frames=N_PER_GPU * T, max_episode_frames=1000)
            col.train_rews = []
            for _ in range(2):
                col.take_actions()                                   # warm the pipes
            t_spawn = time.perf_counter() - t_spawn
            n_col, t0 = 0, time.perf_counter()
            while n_col < T and (time.perf_counter() - t0) < budget_s / 3.0:
                col.take_actions()
                n_col += 1
            collect[procs] = ((time.perf_counter() - t0) / n_col, n_col)
        finally:
            env.close()
        log("cpu baseline: proc_nums=%d: %d collect steps, %.4f s/step (workers up in %.1f s)"
            % (procs, n_col, collect[procs][0], t_spawn))
    # update leg on a full synthetic rollout (values only matter for timing)
    rs = np.random.RandomState(0)
    ring = replay.RingOracle(N_PER_GPU * T, env_nums=N_PER_GPU, time_limit_filter=True)
    ring.data = {"obs": rs.randn(T, N_PER_GPU, D), "next_obs": rs.randn(T, N_PER_GPU, D),
                 "acts": np.tanh(rs.randn(T, N_PER_GPU, A)) * 0.9, "values": rs.randn(T, N_PER_GPU, 1),
                 "rewards": rs.randn(T, N_PER_GPU, 1), "terminals": np.zeros((T, N_PER_GPU, 1)),
                 "time_limits": np.zeros((T, N_PER_GPU, 1))}
    ppo = PPOOracle(pf, ls, vf, entropy_coeff=0.005, opt_epochs=OPT_EPOCHS, batch_size=BATCH_PER_GPU, num_epochs=100000)
    t0 = time.perf_counter()
    ppo.process_epoch_samples(ring)
    t_gae = time.perf_counter() - t0
    keys = ["obs", "acts", "advs", "estimate_returns", "values"]
    n_upd, t0 = 0, time.perf_counter()
    n_mb = N_PER_GPU * T // BATCH_PER_GPU
    while n_upd < OPT_EPOCHS * n_mb and (time.perf_counter() - t0) < budget_s / 3.0:
        for _idx, batch in ring.epoch_minibatches(BATCH_PER_GPU, keys, True):
            ppo.update(batch)
            n_upd += 1
    t_upd = (time.perf_counter() - t0) / n_upd
    log("cpu baseline: %d updates, %.3f s/update" % (n_upd, t_upd))
    steps = N_PER_GPU * T
    full = {p: T * ts + t_gae + OPT_EPOCHS * n_mb * t_upd for p, (ts, _) in collect.items()}
    best = min(full, key=full.get)
    by_procs = {"proc_nums=%d" % p: {"env_steps_per_s": steps / full[p], "collect_env_steps_per_s": N_PER_GPU / collect[p][0],
                                     "collect_steps_timed": collect[p][1]} for p in sorted(collect)}
    return {"value": steps / full[best], "unit": "env-steps/s", "cores": cores, "kind": "port",
            "cpu_model": _cpu_model(), "host_hw_threads": cores, "torch_threads": threads, "env_worker_procs": best,
            "by_proc_nums": by_procs,
            "sample": "collect: %s of %d vector steps of N=%d at proc_nums %s (env worker processes over pipes; %s "
                      "env-steps/s); GAE %.3fs; update: %d of %d minibatch updates of B=%d (%d torch threads, %.3f s each); "
                      "scaled to one full iteration = %s s; value = proc_nums=%d"
                      % ("/".join(str(collect[p][1]) for p in sorted(collect)), T, N_PER_GPU,
                         "/".join(str(p) for p in sorted(collect)),
                         "/".join("%.0f" % (N_PER_GPU / collect[p][0]) for p in sorted(collect)), t_gae, n_upd,
                         OPT_EPOCHS * n_mb, BATCH_PER_GPU, threads, t_upd,
                         "/".join("%.1f" % full[p] for p in sorted(collect)), best)}


def cpu_baseline_full(warm=3, timed=20, procs=4):
    """BASELINE.md section 3, literally: the reference-style CPU path (oracle/, a port) run as WHOLE iterations on this host --
    3 warm-up + 20 timed iterations of {collect T=128 vector steps on N=2048 envs (SubProcVecEnv-like workers, proc_nums =
    4 as the reference's example hard-codes), GAE, 10 epochs x 4 minibatch updates of B=65536} -- reported as median / min /
    max env-steps/s for (a) env-only stepping, (b) collect (policy + env + buffer), (c) the full PPO iteration.  About two
    minutes of host time per ~25 iterations; `python bench.py --cpu-baseline-full` prints one JSON line (kept under profiles/)."""
    import functools
    import torch
    from oracle import nets, replay
    from oracle.collector import VecOnPolicyCollectorOracle
    from oracle.subproc_env import SubProcVecEnvCPU
    from oracle.ppo import PPOOracle
    from oracle.synth_env import SynthSingleEnvCPU
    cores = os.cpu_count() or 1
    threads = min(cores, 32)
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    np.random.seed(0)
    fns = [functools.partial(SynthSingleEnvCPU, i) for i in range(N_PER_GPU)]
    gen = torch.Generator().manual_seed(0)
    pf, vf = nets.init_mlp(D, [H, H], A, generator=gen), nets.init_mlp(D, [H, H], 1, generator=gen)
    ls = torch.full((A,), float(np.log(0.125)))
    env = SubProcVecEnvCPU(procs, N_PER_GPU, fns, SynthSingleEnvCPU(0))
    steps = N_PER_GPU * T
    stat = lambda xs: {"median": float(np.median(xs)), "min": float(np.min(xs)), "max": float(np.max(xs)), "n": len(xs)}
    try:
        ring = replay.RingOracle(steps, env_nums=N_PER_GPU, time_limit_filter=True)
        col = VecOnPolicyCollectorOracle(env, ring, pf, ls, vf, epoch_frames=steps, max_episode_frames=1000)
        ppo = PPOOracle(pf, ls, vf, entropy_coeff=0.005, opt_epochs=OPT_EPOCHS, batch_size=BATCH_PER_GPU, num_epochs=100000)
        keys = ["obs", "acts", "advs", "estimate_returns", "values"]
        t_env, t_col, t_all = [], [], []
        acts = np.zeros((N_PER_GPU, A))
        for it in range(warm + timed):
            t0 = time.perf_counter()
            for _ in range(T):                                              # (a) env-only stepping, zero actions
                env.step(acts)
            t1 = time.perf_counter()
            col.train_one_epoch()                                           # (b) collect
            t2 = time.perf_counter()
            ppo.process_epoch_samples(ring)                                 # GAE
            for _ in range(OPT_EPOCHS):
                for _idx, batch in ring.epoch_minibatches(BATCH_PER_GPU, keys, True):
                    ppo.update(batch)
            t3 = time.perf_counter()
            log("cpu baseline (full): iteration %d: env %.2f s, collect %.2f s, GAE + updates %.2f s" % (it, t1 - t0, t2 - t1, t3 - t2))
            if it >= warm:
                t_env.append(steps / (t1 - t0)); t_col.append(steps / (t2 - t1)); t_all.append(steps / (t3 - t1))
    finally:
        env.close()
    return {"value": float(np.median(t_all)), "unit": "env-steps/s", "cores": cores, "kind": "port", "cpu_model": _cpu_model(),
            "host_hw_threads": cores, "torch_threads": threads, "env_worker_procs": procs,
            "protocol": "BASELINE.md section 3: %d warm-up + %d timed whole iterations; median / min / max" % (warm, timed),
            "env_only_env_steps_per_s": stat(t_env), "collect_env_steps_per_s": stat(t_col), "full_iteration_env_steps_per_s": stat(t_all),
            "sample": "%d whole iterations of N=%d x T=%d, %d x %d updates of B=%d, nothing scaled" % (timed, N_PER_GPU, T, OPT_EPOCHS,
                                                                                                     steps // BATCH_PER_GPU, BATCH_PER_GPU)}


def cpu_baseline_offpolicy(kind):
    """CPU baseline of the secondary workloads (tools/bench_sac.py, tools/bench_dqn.py): the oracle's update on a
    bounded sample (2 timed updates after 1 warm-up) of the same batch shape.  kind: sac | dqn | qrdqn."""
    import numpy as np
    import torch
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    rs = np.random.RandomState(0)
    if kind == "sac":
        from oracle import nets
        from oracle.sac import TwinSACQOracle
        B, H = 4096, 256
        gen = torch.Generator().manual_seed(0)
        o = TwinSACQOracle(nets.init_mlp(17, [H, H], 12, generator=gen), nets.init_mlp(23, [H, H], 1, generator=gen),
                           nets.init_mlp(23, [H, H], 1, generator=gen), w_std=0, w_mean=0)
        batch = {"obs": rs.randn(B, 17), "next_obs": rs.randn(B, 17), "acts": np.tanh(rs.randn(B, 6)),
                 "rewards": rs.randn(B, 1), "terminals": np.zeros((B, 1))}
        step = lambda: o.update(batch, torch.randn(B, 6), torch.randn(B, 6))
        sample = "TwinSACQ update, B=4096, MLP 256x256"
    else:
        from oracle.dqn import DQNOracle
        import torchrl_amd.networks as networks
        from torchrl_amd import ops
        B, A, Q = 512, 6, (200 if kind == "qrdqn" else 1)
        torch.manual_seed(0)
        qf = networks.Net(output_shape=A * Q, base_type=networks.CNNBase, append_hidden_shapes=[512],
                          activation_func=torch.nn.Tanh, input_shape=(4, 84, 84),
                          hidden_shapes=[[16, [8, 8], [4, 4], [0, 0]], [32, [4, 4], [2, 2], [0, 0]], [64, [3, 3], [1, 1], [0, 0]]])
        o = DQNOracle([p.detach().cpu() for p in ops.cnn_param_list(qf)], [4, 2, 1], quantile_num=Q, action_num=A)
        batch = {"obs": rs.randint(0, 256, (B, 4, 84, 84)).astype(np.uint8),
                 "next_obs": rs.randint(0, 256, (B, 4, 84, 84)).astype(np.uint8),
                 "acts": rs.randint(0, A, (B,)), "rewards": rs.randn(B, 1), "terminals": np.zeros((B, 1))}
        step = lambda: o.update(batch)
        sample = "%s update, B=512, conv 16/32/64 + fc512%s" % ("QRDQN" if Q > 1 else "DQN", ", Q=200" if Q > 1 else "")
    step()
    t0 = time.perf_counter()
    step(); step()
    return {"value": 1e3 * (time.perf_counter() - t0) / 2, "unit": "ms/update", "cores": torch.get_num_threads(),
            "kind": "port", "sample": "2 timed " + sample}


def cpu_baseline_subprocess(timeout_s=240):
    """Run the CPU leg in its own interpreter (own torch thread pool, spawn-safe, hard time limit)."""
    import subprocess
    try:
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], timeout=timeout_s,
                             stdout=subprocess.PIPE, stderr=sys.stderr, text=True)
        for line in res.stdout.splitlines():
            if line.startswith("CPU_BASELINE "):
                return json.loads(line[len("CPU_BASELINE "):])
        return {"value": None, "unit": "env-steps/s", "cores": os.cpu_count(), "kind": "port",
                "sample": "failed: rc=%d" % res.returncode}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "env-steps/s", "cores": os.cpu_count(), "kind": "port",
                "sample": "failed: exceeded %ds on this host" % timeout_s}


def _device_map(world):
    """Device index of every local rank.  Default: rank r on GPU r.  TRL_BENCH_DEVICE_MAP="0,0" (tests) places several ranks
    on one device -- RCCL refuses that, so the process group is then gloo and the library's communicator runs without RCCL
    (peer buffers through hipIpc, or torch.distributed with TRL_NO_PEER=1)."""
    spec = os.environ.get("TRL_BENCH_DEVICE_MAP", "").strip()
    if not spec:
        return list(range(world))
    devs = [int(x) for x in spec.split(",")]
    if len(devs) < world:
        raise SystemExit("TRL_BENCH_DEVICE_MAP=%s names %d devices for %d ranks" % (spec, len(devs), world))
    return devs[:world]


def _probe_child():
    """`bench.py --probe-graph-collectives`: a sacrificial process per rank (own rendezvous port) that captures RCCL
    all-reduces of the gradient's size into a HIP graph, replays it and checks the sums.  Exit code 0 = usable."""
    import datetime
    import torch
    import torch.distributed as td
    rank, world, local = (int(os.environ[k]) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"))
    local = _device_map(world)[local]
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    td.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=60))
    n = 11008                                    # the flat [pf | vf] gradient, 44 KB
    src = torch.full((n,), rank + 1.0, device=dev)
    buf = src.clone()
    td.all_reduce(buf)                           # communicator set-up outside the capture
    torch.cuda.synchronize()
    stat = torch.zeros(80, dtype=torch.float64, device=dev)       # the advantage extrema: float64, MAX
    td.all_reduce(stat, op=td.ReduceOp.MAX)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        stat.fill_(float(rank))
        td.all_reduce(stat, op=td.ReduceOp.MAX)
        for _ in range(4):
            buf.mul_(0.5)
            td.all_reduce(buf)
    want = 0.5 * world * (world + 1) / 2 * (0.5 * world) ** 3
    for _ in range(20):
        buf.copy_(src)
        graph.replay()
    torch.cuda.synchronize()
    ok = bool(torch.allclose(buf, torch.full_like(buf, want), rtol=1e-5)) and bool((stat == world - 1.0).all())
    td.destroy_process_group()
    sys.exit(0 if ok else 1)


def probe_graph_collectives(timeout_s=150):
    """True when graph-captured RCCL collectives work on this node, established in child processes so that a hang or
    a crash there costs a time-out, not the benchmark (the children are killed by PID)."""
    import subprocess
    try:
        env = dict(os.environ, MASTER_PORT=str(int(os.environ.get("MASTER_PORT", "29500")) + 17))
        for key in [k for k in env if k.startswith("TORCHELASTIC_")]:  # the children rendezvous on their own TCP store,
            del env[key]                                                # not on the launcher's agent store
        proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--probe-graph-collectives"], env=env,
                                stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:
        return False
    try:
        return proc.wait(timeout=timeout_s) == 0
    except subprocess.TimeoutExpired:
        proc.kill()
        proc.wait()
        return False
    except Exception:                                                   # the probe must never take the benchmark down
        return False


SPAWN_LIMIT_S = 1200.0                            # the whole multi-rank job, children killed by PID afterwards
RANK_GUARD_S = 900.0                              # a rank's own guard against a wedged collective


def self_spawn(n, argv, script=None):
    """`python bench.py --gpus N` without a launcher (no RANK / WORLD_SIZE in the environment): this process becomes the
    launcher -- N children of this same script, one rank per GPU (LOCAL_RANK = rank), a free rendezvous port on
    127.0.0.1 -- relays rank 0's stdout (the ONE JSON line) and exits with the first non-zero child exit code.  A child
    that dies takes the others down (a collective would wait for it forever); nothing outlives SPAWN_LIMIT_S."""
    import socket
    import subprocess
    import threading
    # (the port is free when it is picked, not reserved: a rendezvous that dies within seconds -- somebody else got the port
    # in between, seen once in a test session -- is tried again on another one)
    attempt = int(os.environ.get("TRL_BENCH_SPAWN_ATTEMPT", "0"))
    started = time.time()
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    base = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    log("no launcher environment: spawning %d ranks (rendezvous 127.0.0.1:%d)" % (n, port))
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TRL_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      text=True if r == 0 else None))
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout.read().splitlines()), daemon=True)
    reader.start()
    deadline, rc = time.time() + SPAWN_LIMIT_S, 0
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            rc = next((c for c in codes if c), 0)
            break
        bad = next((c for c in codes if c not in (None, 0)), None)
        if bad is not None or time.time() > deadline:
            rc = bad if bad is not None else 3
            log("rank exit code %s%s: stopping the other ranks" % (rc, "" if bad is not None else " (time limit)"))
            time.sleep(3.0)                                             # let the survivors report their own error first
            for p in procs:
                if p.poll() is None:
                    p.kill()                                            # by PID: exactly the children started above
            break
        time.sleep(0.1)
    for p in procs:
        p.wait()
    reader.join(timeout=10)
    if rc not in (0, 3) and time.time() - started < 20.0 and attempt < 2 and not any(l.startswith("{") for l in lines):
        log("the ranks failed within %.0f s of their start: once more on another port" % (time.time() - started))
        os.environ["TRL_BENCH_SPAWN_ATTEMPT"] = str(attempt + 1)
        return self_spawn(n, argv, script)
    for line in lines:                                                  # the ONE JSON line goes to stdout; anything a library
        print(line, flush=True, file=sys.stdout if line.startswith("{") else sys.stderr)   # printed there (gloo's banner) does not
    return rc


def measure_peaks(dev):
    """the ACHIEVABLE peaks of this box next to the nominal ones -- a 1-GiB 16-byte-access copy kernel
    (HBM), a register-resident fp32 MFMA issue loop (matrix pipe) and a 4096^3 product on the library's own dense-layer
    kernel (what a whole GEMM of its inner loop reaches).  A few hundred milliseconds, after the timed region."""
    import torch
    from torchrl_amd import _C
    lib, stream = _C.lib(), _C.stream_ptr(dev)

    def timed(fn, warm, reps):
        for _ in range(warm):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        b.synchronize()
        return a.elapsed_time(b) * 1e-3 / reps
    out = {"nominal": {"hbm_TBps": 8.0, "f32_mfma_TFLOPs": MFMA_F32_PEAK_TFLOPS}}
    try:
        n = 1 << 28                                                     # 1 GiB of floats
        src, dst = torch.empty(n, device=dev).fill_(1.5), torch.empty(n, device=dev)
        by_mode = {}
        for mode in (0, 1, 2):                                          # three launch shapes of the same 16-byte copy: quote the best
            t = timed(lambda: _C.check(lib.trl_peak_copy_f32(src.data_ptr(), dst.data_ptr(), n, mode, stream), "trl_peak_copy_f32"), 3, 10)
            by_mode["mode%d" % mode] = 2 * 4 * n / t / 1e12
        out["hbm_copy_TBps"] = max(by_mode.values())
        out["hbm_copy_TBps_by_mode"] = by_mode
        del src, dst
        m = 4096
        x, w = torch.randn(m, m, device=dev), torch.randn(m, m, device=dev)
        # (best of three short rounds: right after the sustained MFMA loop the part's clocks are still recovering)
        t = min(timed(lambda: _C.linear_fwd(x, w, None, _C.ACT_NONE), 2, 8) for _ in range(3))
        out["f32_gemm_4096_TFLOPs"] = 2.0 * m * m * m / t / 1e12
        lib_name = "library"
        try:
            lib_name = str(torch.backends.cuda.preferred_blas_library()).split(".")[-1]
        except Exception:                                                # noqa: BLE001
            pass
        wt = w.t()
        t = min(timed(lambda: torch.mm(x, wt), 2, 8) for _ in range(3))  # the vendor SGEMM (calibration leg only, never the product path)
        out["f32_gemm_4096_vendor_TFLOPs"] = 2.0 * m * m * m / t / 1e12
        out["f32_gemm_4096_vendor"] = "torch.mm -> %s" % lib_name
        del x, w, wt
        time.sleep(0.5)                                                 # (the sustained register loop last: what follows it runs on sagging clocks)
        wgs, iters = 1024, 10000
        sink = torch.empty(wgs * 256, device=dev)
        t = timed(lambda: _C.check(lib.trl_peak_mfma_f32(sink.data_ptr(), wgs, iters, stream), "trl_peak_mfma_f32"), 2, 5)
        out["f32_mfma_TFLOPs"] = wgs * 4 * iters * 4 * 4096 / t / 1e12
        # the HBM-bound helpers of the path against the measured copy rate (north_star: HBM GB/s against peak)
        hb = []
        rew, val, term, tl = (torch.rand(T, N_PER_GPU, 1, device=dev) for _ in range(4))
        lastv, adv, ret = torch.rand(N_PER_GPU, 1, device=dev), torch.empty(T, N_PER_GPU, 1, device=dev), torch.empty(T, N_PER_GPU, 1, device=dev)
        t = timed(lambda: _C.gae(rew, val, term, tl, lastv, adv, ret, 0.99, 0.95, True), 5, 50)
        hb.append({"kernel": "gae_scan_kernel (K4, 128 x 2048)", "algorithmic_bytes": 24 * T * N_PER_GPU, "us": t * 1e6})
        frames = torch.zeros(16, 512, 2 * 28224, dtype=torch.uint8, device=dev)   # 16 rows of the cfg 5 ring: {obs, next_obs} stacks
        idxs = [torch.tensor([r], dtype=torch.int64, device=dev) for r in range(16)]    # a different 29 MB row per launch
        dst = torch.empty(1, 512, frames.shape[2], dtype=torch.uint8, device=dev)
        turn = [0]

        def gather_next():
            turn[0] = (turn[0] + 1) % 16
            _C.gather_rows(frames, idxs[turn[0]], out=dst)
        t = timed(gather_next, 5, 48)
        hb.append({"kernel": "gather_rows_kernel u8 (K6, cfg 5: 512 x 2 frame stacks)", "algorithmic_bytes": 2 * dst.numel(), "us": t * 1e6})
        obs = torch.rand(976, 1024, 42, device=dev)
        idx4 = torch.randint(0, 976, (4,), dtype=torch.int64, device=dev)
        dst4 = torch.empty(4, 1024, 42, device=dev)
        t = timed(lambda: _C.gather_rows(obs, idx4, out=dst4), 5, 50)
        hb.append({"kernel": "gather_rows_kernel f32 (K6, cfg 3: 4 rows x 1024 envs x 42 floats)", "algorithmic_bytes": 8 * dst4.numel(), "us": t * 1e6})
        for h in hb:
            h["GBps"] = h["algorithmic_bytes"] / h["us"] / 1e3
            h["frac_of_measured_copy"] = h["GBps"] / 1e3 / out["hbm_copy_TBps"]
            h["frac_of_nominal_8TBps"] = h["GBps"] / 8e3
            h["note"] = "HIP events over back-to-back launches; a few-MB launch is latency-bound (launch + 2-3 dependent memory round trips), not bandwidth-bound"
        out["hbm_kernels"] = hb
        out["how"] = ("hbm: trl_peak_copy_f32, 1 GiB read + 1 GiB written, 10 launches; mfma: trl_peak_mfma_f32, %d workgroups x 4 "
                      "waves x %d x 4 v_mfma_f32_32x32x2_f32 from registers; gemm: trl_linear_fwd_f32 4096^3" % (wgs, iters))
    except Exception as exc:                                            # noqa: BLE001 -- calibration must not cost the line
        out["error"] = repr(exc)
    return out


SECONDARY = (("sac_cfg3", ["tools/bench_sac.py", "--epochs", "10"], 2.59e6 * 4096 / 1e9),
             ("dqn_cfg5", ["tools/bench_dqn.py", "--epochs", "4"], 38e6 * 512 / 1e9),
             ("qrdqn_cfg5", ["tools/bench_dqn.py", "--epochs", "4", "--quantiles", "200"], (4 * 10.9e6 + 12 * 200 * 200) * 512 / 1e9))


def secondary_workloads(timeout_s=150):
    """BASELINE cfg 3 / cfg 5 in the driver-run record: the epochs of tools/bench_{sac,dqn}.py, each in its own interpreter
    (own device memory, a failure costs its entry only), after the headline's timed region.  roofline_frac = the update's
    algorithmic FLOPs / ms_per_update / the nominal fp32 matrix peak."""
    import subprocess
    out = {}
    for name, cmd, gflop in SECONDARY:
        t0 = time.perf_counter()
        try:
            res = subprocess.run([sys.executable, os.path.join(REPO, cmd[0])] + cmd[1:], timeout=timeout_s,
                                 stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            line = next((ln for ln in reversed(res.stdout.splitlines()) if ln.startswith("{")), None)
            if res.returncode != 0 or line is None:
                out[name] = {"error": "rc=%d %s" % (res.returncode, res.stderr[-300:])}
                continue
            d = json.loads(line)
            out[name] = {"ms_per_update": d["ms_per_update"], "updates_per_s": d["updates_per_s"],
                         "env_steps_per_s": d["env_steps_per_s"], "ms_per_vector_step": d["ms_per_vector_step"],
                         "update_gflop": gflop, "roofline_frac": gflop / d["ms_per_update"] / MFMA_F32_PEAK_TFLOPS,
                         "workload": d["workload"]}
        except subprocess.TimeoutExpired:
            out[name] = {"error": "exceeded %ds" % timeout_s}
        except Exception as exc:                                        # noqa: BLE001
            out[name] = {"error": repr(exc)}
        log("secondary %s: %s (%.1f s)" % (name, {k: v for k, v in out[name].items() if k != "workload"}, time.perf_counter() - t0))
    return out


def time_collectives(dist, dev, reps=50):
    """Stand-alone cost of the three exchanges at their real sizes (C1 44 KB gradient SUM, C2 advantage statistics of 40
    minibatches, C3 logging statistics), host-timed over `reps` back-to-back calls with one device wait at the end."""
    import torch
    g = torch.zeros(11085, device=dev)
    raw = torch.zeros(40, 4, dtype=torch.float64, device=dev)
    info = torch.zeros(40, 24, dtype=torch.float64, device=dev)
    out = {}
    for name, fn in (("c1_grad_sum_44KB_us", lambda: dist.all_reduce_sum_(g)), ("c2_adv_stats_us", lambda: dist.reduce_adv_raw_(raw)),
                     ("c3_info_stats_us", lambda: dist.reduce_info_(info))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        torch.distributed.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        out[name] = 1e6 * (time.perf_counter() - t0) / reps
    return out


PARITY_SLACK = 1.15                               # headline = the reference's noise stream when it costs at most this factor


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--transport", default="auto", choices=["auto", "peer", "rccl"],
                    help="multi-rank exchange of the 44 KB gradient / statistics: peer = xGMI-mapped buffers inside the "
                         "fold launch (fail if the self-check does not pass), rccl = all-reduce calls, auto = peer when its "
                         "self-check passes on every rank, else rccl")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the cfg 3 / cfg 5 epochs and the peak calibration")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="BASELINE.md section 3 literally: 3 warm-up + 20 timed whole CPU iterations (median / min / max); "
                         "prints one JSON line and exits (about two minutes, no GPU work)")
    ap.add_argument("--workload", default="ppo", choices=["ppo", "sac", "dqn", "qrdqn"],
                    help="with --cpu-baseline-only: which workload's CPU baseline to time (tools/bench_{sac,dqn}.py)")
    ap.add_argument("--probe-graph-collectives", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if os.environ.get("TRL_BENCH_DUMP_AFTER_S") and ("RANK" in os.environ or args.gpus == 1):   # diagnosing a wedged rank: every thread's stack, then exit
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["TRL_BENCH_DUMP_AFTER_S"]), exit=True)
    if args.probe_graph_collectives:
        _probe_child()
    if args.cpu_baseline_full:
        print(json.dumps({"cpu_baseline_full": cpu_baseline_full()}), flush=True)
        return
    if args.cpu_baseline_only:                      # child mode of the cpu_baseline leg
        res = cpu_baseline() if args.workload == "ppo" else cpu_baseline_offpolicy(args.workload)
        print("CPU_BASELINE " + json.dumps(res), flush=True)
        return
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(self_spawn(args.gpus, sys.argv[1:]))   # plain `python bench.py --gpus N`: be the launcher
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and "RANK" in os.environ:
        log("WORLD_SIZE=%d from the launcher overrides --gpus %d" % (world, args.gpus))
    devmap = _device_map(max(world, local + 1))
    local_dev = devmap[local]
    if local_dev >= torch.cuda.device_count():
        raise SystemExit("rank %d wants cuda:%d but %d device(s) are visible" % (rank, local_dev, torch.cuda.device_count()))
    backend, comm_info = None, {}
    forced = os.environ.get("TRL_FORCE_COLLECTIVES") == "1" and "RANK" in os.environ   # 1-GPU smoke test of the RCCL path
    if world > 1 or forced:
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local_dev)
        shared = len(set(devmap[:world])) < world                         # several ranks per device (tests): no RCCL
        backend = "gloo" if shared else "nccl"
        if shared:
            td.init_process_group("gloo", rank=rank, world_size=world)
        else:
            td.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_dev))
        import threading
        guard = threading.Timer(RANK_GUARD_S, lambda: os._exit(3))           # a wedged collective must not outlive the run
        guard.daemon = True
        guard.start()
        # The library's own communicator (include/trl_hip.h, trl_comm_*): RCCL for bandwidth-class messages plus
        # peer-mapped xGMI buffers for the 44 KB gradient and the statistics vectors.  With the peers up (self-checked
        # on every rank) the multi-rank sequence consists of plain kernel launches and is graph-replayed like the
        # single-process one.  Otherwise: RCCL all-reduces, captured into the graph only after child processes have
        # shown that graph-captured collectives work on this node (TRL_GRAPH_COLLECTIVES overrides the probe).
        from torchrl_amd import dist as _dist
        # Per-rank CPU slice before any thread pool exists (the reference-noise draw runs a pool of host threads per rank,
        # one rollout ahead of the device; TRL_RANK_AFFINITY=0 leaves the scheduler alone).
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        if os.environ.get("TRL_RANK_AFFINITY", "1") != "0":
            mine = _dist.pin_rank_cpus(local, local_world)
            comm_info["cpu_affinity"] = ("%d CPUs per rank (rank 0: %d-%d)" % (len(mine), mine[0], mine[-1])) if mine and rank == 0 \
                else ("%d CPUs per rank" % len(mine) if mine else "not pinned")
        # Pre-flight (rank 0 reports): can every pair of this job's GPUs reach each other, and over what
        if not shared and torch.cuda.device_count() >= world and rank == 0:   # (one process asks: queries only, no contexts on the other GPUs)
            try:
                comm_info["link_preflight"] = _dist.link_preflight(devmap[:world])
            except Exception as exc:                                      # noqa: BLE001 -- a diagnostic must not cost the run
                comm_info["link_preflight"] = {"error": repr(exc)}
        elif rank == 0:
            comm_info["link_preflight"] = {"note": "%d ranks on %d device(s): nothing to probe" % (world, len(set(devmap[:world])))}
        if args.transport == "rccl":
            os.environ["TRL_NO_PEER"] = "1"
        peers = _dist.init_comm(torch.device("cuda", local_dev))
        report = _dist.peer_report()
        comm_info["transport_requested"] = args.transport
        comm_info["peer_self_check"] = "passed on every rank" if peers else \
            ("disabled (--transport rccl / TRL_NO_PEER=1)" if os.environ.get("TRL_NO_PEER") == "1"
             else report.get("self_check", "failed or unavailable"))
        if report.get("not_used"):
            comm_info["peer_transport_not_used"] = report["not_used"]
        comm_info["ranks_per_device"] = report.get("ranks_per_device", 1 if not shared else None)
        for key in ("peer_buffer", "wait_footprint"):                      # which allocation the peer buffer got; resident
            if report.get(key):                                            # footprint of a fold launch that waits for other ranks
                comm_info[key] = report[key]
        comm_info["transport_vote"] = "peer" if peers else ("all-reduce calls" + (" (peer self-check %s)" % report["self_check"]
                                                                                   if report.get("self_check") else ""))
        if args.transport == "peer" and not peers:
            raise SystemExit("--transport peer: the peer-mapped transport did not pass its self-check on every rank")
        log("peer transport: %s" % ("up (self-check passed on every rank)" if peers else "unavailable -> all-reduce calls"))
        if not peers and "TRL_GRAPH_COLLECTIVES" not in os.environ:
            if backend == "nccl":
                probed = probe_graph_collectives()
                flag = torch.tensor([1.0 if probed else 0.0], device=torch.device("cuda", local_dev))
                td.all_reduce(flag, op=td.ReduceOp.MIN)                    # every rank takes the same route
                os.environ["TRL_GRAPH_COLLECTIVES"] = "1" if flag.item() == 1.0 else "0"
                log("graph-captured RCCL collectives: %s" % ("on" if flag.item() == 1.0 else "off (probe failed)"))
            else:
                os.environ["TRL_GRAPH_COLLECTIVES"] = "0"                  # host-staged gloo calls cannot be captured
    dev = torch.device("cuda", local_dev)
    torch.cuda.set_device(dev)

    from torchrl_amd import dist
    log("building agent on %s" % dev)
    agent, col = build_agent(dev, world, rank)
    col.env.reset()
    eng = agent.engine()
    torch.cuda.synchronize()
    # The engine runs a launch sequence from the stream the first time it sees a shape, captures it into a HIP graph
    # the second time and replays it from the third: with fewer than 3 warm-up steps the capture is done here, as
    # set-up, so that the timed region always measures the steady state.
    setup = max(0, 3 - args.warmup) if os.environ.get("TRL_NO_GRAPH") != "1" else 0
    epoch = [0]

    def run_iterations(n, sync_each=False, note=None):
        for _ in range(n):
            iteration(agent, col, epoch[0])
            epoch[0] += 1
            if sync_each:
                torch.cuda.synchronize()
                if note:
                    log(note % (epoch[0] - 1))

    if world > 1 and dist.peer_ready():
        # Insurance for the peer transport: its first three iterations (stream launches, graph capture, first replay)
        # run guarded.  A rank whose bounded waits trip raises; every rank then votes, and on any failure ALL ranks drop
        # the peer buffers and continue on all-reduce calls with a fresh agent, instead of losing the run.
        import torch.distributed as td
        ok = 1.0
        try:
            for _ in range(3):
                run_iterations(1)
                agent.logger.drain()                                      # (check_comm runs where the statistics are read)
                torch.cuda.synchronize()
        except Exception as exc:                                          # noqa: BLE001 -- anything: fall back, loudly
            log("peer transport failed in the first iterations on rank %d (iteration %d): %r; comm error word %s"
                % (rank, epoch[0], exc, dist.comm_error_detail()))
            ok = 0.0
        vote = torch.tensor([ok], device=dev)
        td.all_reduce(vote, op=td.ReduceOp.MIN)
        if vote.item() != 1.0 and args.transport == "peer":
            raise SystemExit("--transport peer: a guarded iteration failed on some rank (see the rank logs)")
        if vote.item() != 1.0:
            log("falling back to all-reduce calls on every rank")
            comm_info["guarded_iterations"] = "failed on some rank -> fell back to all-reduce calls"
            dist.destroy_comm()
            dist.init_comm(dev, peers=False)
            os.environ.setdefault("TRL_GRAPH_COLLECTIVES", "0")
            agent, col = build_agent(dev, world, rank)
            col.env.reset()
            eng = agent.engine()
            epoch[0] = 0
            torch.cuda.synchronize()
        else:
            comm_info["guarded_iterations"] = "3 completed on every rank"
            log("peer transport: three guarded iterations completed on every rank")
    run_iterations(setup, True, "set-up iteration %d done (graph capture)")
    # A full (generation 2) collection of the interpreter's heap costs tens of milliseconds with torch loaded -- as long as the
    # whole timed region -- and when one falls due is a matter of allocation counts: the objects alive after set-up are
    # moved out of the collector's reach so that the timed region measures the loop, not the dice (seen in
    # tools/bench_sac.py as a 70 ms hole in one of two timed loops, at random).  Before the warm-up, which absorbs the
    # collection's own after-effects.
    import gc
    gc.collect()
    gc.freeze()
    log("warmup x%d" % args.warmup)
    run_iterations(args.warmup, True, "warmup iteration %d done")

    def timed_region():
        """EXACTLY `--steps` iterations between {barrier, device wait} pairs; every iteration's update statistics are read
        inside (the last ones before the clock stops).  Returns (seconds, per-iteration marks, info dicts read)."""
        if dist.initialized():
            torch.distributed.barrier()
        torch.cuda.synchronize()
        agent.logger.drain()
        agent.logger.updates = 0
        t0 = time.perf_counter()
        marks = [t0]
        for _ in range(args.steps):
            run_iterations(1)
            marks.append(time.perf_counter())
        agent.logger.drain()                                               # the last iteration's update statistics
        read = agent.logger.updates
        torch.cuda.synchronize()
        if dist.initialized():
            torch.distributed.barrier()
        return time.perf_counter() - t0, marks, read

    # HIP events around every launch of the dominant kernel, on the stream it is launched on.  One process: the
    # minibatch loop of the timed region replays a captured HIP graph, whose nodes cannot be bracketed by
    # events, so the same kernel on the same data is timed in a follow-up pass right after the timed region.
    graph_mode = (not dist.collectives_active() or dist.peer_ready() or os.environ.get("TRL_GRAPH_COLLECTIVES") == "1") \
        and os.environ.get("TRL_NO_GRAPH") != "1"
    probes = []
    eng.probe = None if graph_mode else probes
    elapsed, marks, infos_read = timed_region()
    log("device-noise mode: timed %d iterations in %.3f s" % (args.steps, elapsed))
    log("per-iteration ms: " + " ".join("%.2f" % (1e3 * (b - a)) for a, b in zip(marks[:-1], marks[1:])))
    if graph_mode:
        eng.probe = probes
        run_iterations(PROBE_STEPS)
        torch.cuda.synchronize()
    eng.probe = None
    # The reference's exploration-noise stream (CPU torch generator, distribution.py:60-76 -- what the parity tests run)
    # at one rank: the same iteration with the NEXT rollout's block drawn by a host thread while the device works
    # (collector/on_policy.py::_NoisePrefetcher; bit-identical buffers and parameters to the in-place draws,
    # tests/test_noise_prefetch_gpu.py).  Timed with the same protocol; it is the headline when it costs <= PARITY_SLACK.
    device_elapsed, parity_elapsed = elapsed, None
    col.noise_mode, col.prefetch_noise = "host", True
    can_parity = col._can_prefetch(T)                                   # this rank's rows of every step's draw, from derived states
    col.noise_mode, col.prefetch_noise = "device", False
    if dist.initialized():                                              # every rank takes the same route
        flag = torch.tensor([1.0 if can_parity else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce_max_(flag.neg_())
        can_parity = float(flag.item()) == -1.0
    if can_parity:
        col.noise_mode, col.prefetch_noise = "host", True
        run_iterations(5, True)                                            # (first block drawn in place, pipeline primed)
        gc.collect()
        gc.freeze()                                                        # (as before the first timed region)
        parity_elapsed, pmarks, pread = timed_region()
        log("reference-noise mode: timed %d iterations in %.3f s" % (args.steps, parity_elapsed))
        pre = getattr(col, "_prefetcher", None)
        if pre is not None:                                                # how the blocks reached the device (all rollouts so far)
            from torchrl_amd.collector import noise as _noise
            comm_info["noise_blocks"] = dict(pre.transport_counts, dropped=pre.dropped_blocks,
                                             host_threads=pre.draw_threads or _noise.default_threads(),
                                             jump_ahead_passes=_noise.STATS["jump_passes"],
                                             native_helper=_noise.native_helper() is not None)
        log("per-iteration ms: " + " ".join("%.2f" % (1e3 * (b - a)) for a, b in zip(pmarks[:-1], pmarks[1:])))
        col.stop_noise_prefetch()
        col.noise_mode, col.prefetch_noise = "device", False
    if dist.initialized():                                              # MAX over ranks of both timings, then ONE decision
        tmax = torch.tensor([device_elapsed, parity_elapsed if parity_elapsed is not None else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce_max_(tmax)
        device_elapsed = float(tmax[0].item())
        parity_elapsed = float(tmax[1].item()) if parity_elapsed is not None else None
    elapsed = device_elapsed
    headline_parity = parity_elapsed is not None and parity_elapsed <= PARITY_SLACK * device_elapsed
    if headline_parity:
        elapsed, infos_read = parity_elapsed, pread
    coll_us = time_collectives(dist, dev) if (dist.initialized() and dist.collectives_active()) else None

    if rank != 0:
        _shutdown_dist()
        return
    env_steps = world * N_PER_GPU * T * args.steps
    grad_ms = [s.elapsed_time(e) for s, e in probes]
    avg_s = float(np.mean(grad_ms)) * 1e-3 if grad_ms else float("nan")
    flops = FLOP_PER_SAMPLE * BATCH_PER_GPU
    achieved = flops / avg_s / 1e12
    transport = dist.transport() if dist.collectives_active() else "none (one rank)"
    out = {
        "metric": "env_steps_per_sec_ppo_2048env_halfcheetah_shape",
        "value": env_steps / elapsed, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "PPO cfg2: %d synthetic envs/GPU (17-d obs, 6-d act) x %d-step rollout, GAE, "
                               "%d opt epochs x %d minibatches of %d, MLP 17-64-64-{6,1} tanh"
                               % (N_PER_GPU, T, OPT_EPOCHS, N_PER_GPU * T // BATCH_PER_GPU, BATCH_PER_GPU),
                   "envs_per_gpu": N_PER_GPU, "rollout_steps": T, "batch_per_gpu": BATCH_PER_GPU,
                   "opt_epochs": OPT_EPOCHS,
                   "exploration_noise": ("CPU torch generator = the reference's stream (bit-parity configuration), the next "
                                         "rollout's block drawn by host threads while the device works"
                                         + ("" if world == 1 else "; every rank draws only its rows of each step's (N_total, A) "
                                            "tensor, from the engine state at their position in the stream")) if headline_parity
                   else "device Philox4x32-10 keyed by the global env index",
                   "update_launch_sequences": ("two: the critic's and the actor's updates (ppo.py:93-122 / 41-91) as two concurrent "
                                               "launch sequences on two streams (152 + 104 workgroups per gradient launch), the next "
                                               "rollout behind the policy's, its value pass behind the value function's; the probe pass "
                                               "times the pair as ONE 256-workgroup launch (roofline.avg_launch_us)")
                   + ("" if world == 1 else "; each sequence's fold launch carries its network's gradient SUM over ranks")
                   if getattr(eng, "two_chains", False) and (not dist.collectives_active() or eng.chains_across_ranks())
                   else "joint (one sequence, both networks per gradient launch)",
                   "setup_iterations": setup,
                   "update_infos_read_in_timed_region": infos_read,
                   "host_pipeline": "the info dicts of iteration i's updates are read while iteration i+1's rollout runs "
                                    "(the last ones before the clock stops); `algo.eager_update_infos = True` reads them in place",
                   "transport": transport,
                   "parallelism": "env-sharded dp%d, gradient SUM %s" % (
                       world, "inside the fold/clip/Adam launch over peer-mapped xGMI buffers" if dist.peer_ready()
                       else ("by all-reduce calls (%s)" % transport if dist.collectives_active() else "not needed (one rank)"))},
        "roofline": {"bound": "mfma", "kernel": "ppo_grad_wave_kernel<17,64,6,tanh>", "achieved": achieved,
                     "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_F32_PEAK_TFLOPS,
                     "traffic": pmc_traffic()[0], "traffic_stamp": pmc_traffic()[1],
                     # traffic / mfma_busy are rocprofv3 --pmc figures of committed passes over THIS tree's kernel (the
                     # digest stamp says so), not values of this process: counters cannot be collected inside the timed run
                     "traffic_measured_in_this_run": False, "mfma_busy_measured_in_this_run": False,
                     "flop_per_launch": flops, "avg_launch_us": avg_s * 1e6,
                     "whole_iteration_frac": FLOP_PER_ENV_STEP * env_steps / elapsed / world / 1e12 / MFMA_F32_PEAK_TFLOPS,
                     # the HBM side of the same iteration (north_star: fraction of the HBM roofline): SURVEY.md 8(d)'s 1 240
                     # algorithmic bytes per env-step against the nominal 8 TB/s -- the path is MFMA-bound, this says by how much
                     "hbm_whole_iteration_frac": BYTES_PER_ENV_STEP * env_steps / elapsed / world / 8e12,
                     "hbm_whole_iteration_GBps": BYTES_PER_ENV_STEP * env_steps / elapsed / world / 1e9,
                     "mfma_busy": pmc_traffic()[2],
                     "launches_timed": len(grad_ms),
                     "timed_in": ("follow-up pass of %d iterations (the timed region replays a HIP graph)" % PROBE_STEPS)
                     if graph_mode else "the timed region"},
    }
    if backend is None and comm_info:
        out["config"].update(comm_info)                                    # (one rank: how the noise blocks travelled)
    if backend is not None:
        out["config"]["process_group_backend"] = backend
        out["config"].update(comm_info)
        out["config"]["collective_us"] = coll_us
        n_mb = OPT_EPOCHS * (N_PER_GPU * T // BATCH_PER_GPU)
        out["config"]["exchanges_per_iteration"] = {
            "c1_gradient_sum_44KB": n_mb, "c2_advantage_statistics": OPT_EPOCHS, "c3_logging_statistics": 1,
            "note": "C1 inside the fold/clip/Adam launch on the peer transport; stand-alone costs in collective_us -- "
                    "the 1 -> N curve decomposes as iteration(1 GPU) + %d x C1 + %d x C2 + C3 + time-skew waits" % (n_mb, OPT_EPOCHS)}
        if os.environ.get("TRL_BENCH_SPAWNED") == "1":
            out["config"]["launcher"] = "bench.py spawned its own ranks"
    if parity_elapsed is not None:
        out["device_noise_ms_per_step"] = 1e3 * device_elapsed / args.steps
        out["parity_mode_ms_per_step"] = 1e3 * parity_elapsed / args.steps
        out["config"]["headline_mode"] = "parity" if headline_parity else \
            "device (the reference-noise mode cost more than %.2fx on this host)" % PARITY_SLACK
        out["config"]["parity_mode"] = ("exploration noise from the CPU torch generator (reference stream, prefetched one rollout "
                                        "ahead), %d iterations timed like the headline" % args.steps)
    if world == 1 and not args.no_secondary:
        out["peaks_measured"] = measure_peaks(dev)
        mf = out["peaks_measured"].get("f32_mfma_TFLOPs")
        if mf:
            out["roofline"]["frac_of_measured_peak"] = achieved / mf
        if out["peaks_measured"].get("hbm_kernels"):
            out["roofline"]["hbm"] = out["peaks_measured"]["hbm_kernels"]
        out["secondary"] = secondary_workloads()
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_subprocess()
    print(json.dumps(out))
    _shutdown_dist()


def _shutdown_dist():
    import torch.distributed as td
    from torchrl_amd import dist as _dist
    _dist.destroy_comm()
    if td.is_available() and td.is_initialized():
        td.destroy_process_group()


TRAFFIC_FILE = os.path.join(REPO, "profiles", "grad_kernel_traffic.json")
TRAFFIC_SOURCES = ("torchrl_amd/csrc/k_ppo.hip", "torchrl_amd/csrc/trl_mlp.h")   # what the dominant kernel is compiled from


def kernel_source_digest(root=REPO):
    """sha256 over the dominant kernel's sources: the stamp that ties a committed PMC measurement to the code it measured
    (the GPU box has no .git, so a commit id cannot be checked there; the digest can)."""
    import hashlib
    h = hashlib.sha256()
    for rel in TRAFFIC_SOURCES:
        with open(os.path.join(root, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def pmc_traffic(path=None, root=REPO):
    """(bytes per launch | None, stamp, MFMA-busy record | None): HBM bytes per launch of the dominant kernel from the committed PMC passes
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE cannot be collected from inside the timed run; the JSON records the
    measurement, its gfx950 correction, the source CSV, the commit and the digest of the kernel sources it was taken at --
    tools/stamp_traffic.py writes it).  The value is reported ONLY while the kernel sources are the ones that were measured:
    after any edit of k_ppo.hip / trl_mlp.h it is null until the counters are collected again."""
    try:
        with open(path or TRAFFIC_FILE) as f:
            rec = json.load(f)
        value = rec["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None, {"status": "no committed measurement"}, None
    stamp = {"measured_at_commit": rec.get("measured_at_commit"), "kernel_source_sha256": rec.get("kernel_source_sha256"),
             "source": rec.get("source")}
    try:
        now = kernel_source_digest(root)
    except OSError:
        now = None
    stamp["tree_matches"] = bool(now) and now == rec.get("kernel_source_sha256")
    if not stamp["tree_matches"]:
        stamp["status"] = "stale: the kernel sources changed since the counters were collected -> traffic = null"
        return None, stamp, None
    stamp["status"] = "current"
    # matrix-pipe occupancy of the same kernel from the SQ counter pass of the same tree (rocprofv3 --pmc, a run of its own):
    # SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 32) = share of the launch's SIMD-cycles with an MFMA in the pipe
    return value, stamp, rec.get("mfma_busy")


if __name__ == "__main__":
    main()
