#!/usr/bin/env python
"""Build-container check (last row): is the CPU baseline that bench.py times -- the oracle's
port of the reference's CPU path -- as fast as the REFERENCE ITSELF on the same cores?

Times, side by side on this host, the two legs bench.py's `cpu_baseline` reports:
  collect  the reference's SubProcVecEnv + VecOnPolicyCollector + OnPolicyReplayBuffer   vs
           oracle.subproc_env.SubProcVecEnvCPU + VecOnPolicyCollectorOracle + RingOracle,
           same worker count, same per-env Python objects (oracle.synth_env.SynthSingleEnvCPU), N = 2048;
  update   the reference's PPO.update   vs   oracle.ppo.PPOOracle.update   on the same B = 65 536 minibatch,
           same torch thread count.
Needs /root/reference (imported through the stand-in gym / toolz / cv2 modules of tests/golden/make_golden.py), so it
runs in the build container only -- like make_golden.py next to it, a checker of the checker; no test, bench.py or
smoke() runs it.  Prints one JSON line.

    python tests/golden/compare_cpu_baseline.py [--procs 4] [--steps 12] [--updates 3]
"""
import argparse
import functools
import importlib.util
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N, T, B, D, A, H = 2048, 128, 65536, 17, 6, 64


def load_generator():
    spec = importlib.util.spec_from_file_location("_make_golden", os.path.join(REPO, "tests", "golden", "make_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    return gen


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=4)       # examples/ppo_continuous_vec_subproc.py:31-36 hard-codes 4
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--updates", type=int, default=3)
    args = ap.parse_args()
    gen = load_generator()
    gen.install_stubs()
    # spawned env workers of the reference re-import their modules: give them the same path
    os.environ["PYTHONPATH"] = os.pathsep.join(p for p in sys.path if p)
    import gym
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.collector.on_policy import VecOnPolicyCollector
    from torchrl.env.subproc_vecenv import SubProcVecEnv
    from torchrl.replay_buffers.on_policy import OnPolicyReplayBuffer
    from oracle import nets, replay
    from oracle.collector import VecOnPolicyCollectorOracle
    from oracle.ppo import PPOOracle
    from oracle.subproc_env import SubProcVecEnvCPU
    from oracle.synth_env import SynthSingleEnvCPU
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    net = dict(hidden_shapes=[H, H], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.Tanh)
    pf = policies.GuassianContPolicyBasicBias(input_shape=D, output_shape=A, tanh_action=True, **net)
    vf = networks.Net(input_shape=(D,), output_shape=1, **net)

    # ---- collect leg: the reference ----
    # one (env_func, env_args) pair for all envs: the list form trips `assert len(env_args) == env_args` (vecenv.py:19);
    # every env then has seed 0, which does not matter for the timing
    env = SubProcVecEnv(args.procs, N, SynthSingleEnvCPU, (0,))
    try:
        buf = OnPolicyReplayBuffer(N * T, env_nums=N, time_limit_filter=True)
        col = VecOnPolicyCollector(vf, env=env, eval_env=env, pf=pf, replay_buffer=buf, device=torch.device("cpu"),
                                   train_render=False, epoch_frames=N * T, max_episode_frames=1000, eval_episodes=1)
        col.train_rews = []
        for _ in range(2):
            col.take_actions()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            col.take_actions()
        ref_step = (time.perf_counter() - t0) / args.steps
    finally:
        env.close()

    # ---- collect leg: the port ----
    linears = lambda m: [p.detach().clone() for layer in m.modules() if isinstance(layer, torch.nn.Linear)
                         for p in (layer.weight, layer.bias)]
    pf_p, vf_p = linears(pf), linears(vf)
    ls = pf.logstd.detach().clone()
    oenv = SubProcVecEnvCPU(args.procs, N, [functools.partial(SynthSingleEnvCPU, i) for i in range(N)], SynthSingleEnvCPU(0))
    try:
        ring = replay.RingOracle(N * T, env_nums=N, time_limit_filter=True)
        ocol = VecOnPolicyCollectorOracle(oenv, ring, pf_p, ls, vf_p, epoch_frames=N * T, max_episode_frames=1000)
        ocol.train_rews = []
        for _ in range(2):
            ocol.take_actions()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ocol.take_actions()
        port_step = (time.perf_counter() - t0) / args.steps
    finally:
        oenv.close()

    # ---- update leg ----
    rs = np.random.RandomState(0)
    batch = {"obs": rs.randn(B, D), "acts": np.tanh(rs.randn(B, A)) * 0.9, "advs": rs.randn(B, 1),
             "estimate_returns": rs.randn(B, 1), "values": rs.randn(B, 1)}
    stub_env = type("E", (), {"action_space": gym.spaces.Box(-1, 1, (A,)), "observation_space": gym.spaces.Box(-1, 1, (D,))})()
    agent = gen.make_ppo(pf, vf, stub_env, None, gen._StubCollector(), gen.NullLogger(), batch_size=B, opt_epochs=10)
    agent.update(batch)
    t0 = time.perf_counter()
    for _ in range(args.updates):
        agent.update(batch)
    ref_upd = (time.perf_counter() - t0) / args.updates
    o = PPOOracle(pf_p, ls, vf_p, entropy_coeff=0.005, opt_epochs=10, batch_size=B, num_epochs=100000)
    o.update(batch)
    t0 = time.perf_counter()
    for _ in range(args.updates):
        o.update(batch)
    port_upd = (time.perf_counter() - t0) / args.updates

    full = lambda step, upd: N * T / (T * step + 40 * upd)
    print(json.dumps({
        "host_cores": os.cpu_count(), "env_worker_processes": args.procs, "torch_threads": threads,
        "collect_s_per_vector_step": {"reference": ref_step, "port": port_step, "port_over_reference": port_step / ref_step},
        "update_s_per_minibatch": {"reference": ref_upd, "port": port_upd, "port_over_reference": port_upd / ref_upd},
        "env_steps_per_s_full_iteration": {"reference": full(ref_step, ref_upd), "port": full(port_step, port_upd)},
        "sample": "%d vector steps of N=%d; %d updates of B=%d" % (args.steps, N, args.updates, B)}))


if __name__ == "__main__":
    main()
