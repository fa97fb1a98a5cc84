"""Secondary measurement: Twin-Q SAC at BASELINE cfg 3 (1024 envs, 1e6-transition replay = 976 rows,
B = 4096, MLP 256x256 ReLU): per epoch 16 vector steps (16 384 env-steps) + 16 updates.
Prints one JSON line; `--cpu` adds bench.py's CPU baseline of the same update (bounded sample: 2 updates)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

N, ROWS, B, H, STEPS, OPT = 1024, 976, 4096, 256, 16, 16


class CountingLog:
    """Takes the info dicts as torchrl_amd.utils.Logger does between two rows (now, or `later` for updates that were
    launched but not waited for) and counts them; `drain()` reads what is outstanding."""
    def __init__(self): self.updates, self._later = 0, []
    def add_update_info(self, d): self.drain(); self.updates += 1
    def add_update_infos_later(self, resolve): self._later.append(resolve)
    def drain(self):
        later, self._later = self._later, []
        for resolve in later:
            self.updates += len(resolve())
    def add_epoch_info(self, *a, **k): pass
    def log(self, *a): pass
    def finish(self): pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=20)
    ap.add_argument("--cpu", action="store_true")
    args = ap.parse_args()
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.algo import TwinSACQ
    from torchrl.collector import VecCollector
    from torchrl.env import get_vec_env
    from torchrl.replay_buffers import BaseReplayBuffer
    dev = torch.device("cuda:0")
    torch.manual_seed(0); np.random.seed(0)
    net = dict(hidden_shapes=[H, H], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.ReLU)
    pf = policies.GuassianContPolicy(input_shape=17, output_shape=12, tanh_action=True, **net)
    qf1 = networks.QNet(input_shape=23, output_shape=1, **net)
    qf2 = networks.QNet(input_shape=23, output_shape=1, **net)
    env = get_vec_env("SynthHalfCheetah-v0", {"reward_scale": 1, "obs_norm": False}, N)
    eval_env = get_vec_env("SynthHalfCheetah-v0", {"reward_scale": 1, "obs_norm": False}, N)
    buf = BaseReplayBuffer(ROWS * N, env_nums=N)
    col = VecCollector(env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=dev, epoch_frames=N * STEPS,
                       max_episode_frames=999, noise_mode="device")
    agent = TwinSACQ(pf=pf, qf1=qf1, qf2=qf2, plr=3e-4, qlr=3e-4, policy_std_reg_weight=0, policy_mean_reg_weight=0,
                     automatic_entropy_tuning=True, noise_mode="device", env=env, replay_buffer=buf, collector=col,
                     logger=CountingLog(), discount=0.99, num_epochs=1, batch_size=B, device=dev, save_dir=None, tau=0.005,
                     opt_times=OPT)
    log = agent.logger
    for _ in range(3):
        col.rollout(STEPS); agent.update_per_epoch()
    log.drain(); torch.cuda.synchronize()
    import gc
    gc.collect(); gc.freeze()       # (a generation-2 collection costs ~70 ms with torch loaded: keep it out of either loop)
    # (1) the two phases timed apart: a host wait after each (the update window includes reading its info dicts)
    tc = tu = 0.0
    for _ in range(args.epochs):
        a = time.perf_counter(); col.rollout(STEPS); torch.cuda.synchronize()
        b = time.perf_counter(); agent.update_per_epoch(); log.drain(); torch.cuda.synchronize()
        tc += b - a; tu += time.perf_counter() - b
    # (2) whole epochs in RLAlgo.train's order (rl_algo.py:111-118): collect, update, then look at the collector's result;
    # the update's info dicts are read when the logger would need them (here: one epoch later), nothing else waits
    log.updates = 0
    t0 = time.perf_counter(); seen = 0
    for _ in range(args.epochs):
        res = col.train_one_epoch()
        log.drain()
        agent.update_per_epoch()
        seen += len(res["train_rewards"])
    log.drain(); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    assert log.updates == args.epochs * OPT
    out = {"workload": "TwinSACQ cfg3: %d envs, %d-row replay, B=%d, MLP %dx%d relu, %d steps + %d updates / epoch"
                       % (N, ROWS, B, H, H, STEPS, OPT),
           "env_steps_per_s": args.epochs * N * STEPS / el,
           "env_steps_per_s_phases_timed_apart": args.epochs * N * STEPS / (tc + tu), "updates_per_s": args.epochs * OPT / tu,
           "ms_per_update": 1e3 * tu / (args.epochs * OPT), "ms_per_vector_step": 1e3 * tc / (args.epochs * STEPS)}
    if args.cpu:
        base = cpu_baseline_via_bench("sac")
        out["cpu_oracle_ms_per_update"], out["cpu_threads"] = base["value"], base["cores"]
    print(json.dumps(out))


def cpu_baseline_via_bench(workload):
    """The oracle-based CPU baseline lives in bench.py's cpu_baseline leg (the only non-test place that may touch
    oracle/); this runs it as a subprocess and returns its JSON."""
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--cpu-baseline-only", "--workload", workload],
                         capture_output=True, text=True, timeout=900)
    for line in res.stdout.splitlines():
        if line.startswith("CPU_BASELINE "):
            return json.loads(line[len("CPU_BASELINE "):])
    raise RuntimeError("cpu baseline failed: " + res.stderr[-500:])


if __name__ == "__main__":
    main()
