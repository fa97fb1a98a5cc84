"""Secondary measurement: DQN / QR-DQN at BASELINE cfg 5 (512 envs, 84x84x4 uint8 frames, 1e5-transition
replay = 195 rows, conv net of config/dqn_pong.json, B = 512): per epoch 8 vector steps + 8 updates."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
N, ROWS, B, STEPS, OPT, A = 512, 195, 512, 8, 8, 6
CONVS = [[16, [8, 8], [4, 4], [0, 0]], [32, [4, 4], [2, 2], [0, 0]], [64, [3, 3], [1, 1], [0, 0]]]


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


def run(Q, epochs, cpu, dedup=False):
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.algo import DQN, QRDQN
    from torchrl.collector import VecCollector
    from torchrl.env import get_vec_env
    from torchrl.replay_buffers import BaseReplayBuffer, MemoryEfficientReplayBuffer
    dev = torch.device("cuda:0")
    torch.manual_seed(0); np.random.seed(0)
    qf = networks.Net(output_shape=A * Q, base_type=networks.CNNBase, append_hidden_shapes=[512],
                      activation_func=torch.nn.Tanh, input_shape=(4, 84, 84), hidden_shapes=CONVS)
    env = get_vec_env("SynthAtari-v0", {"reward_scale": 1}, N)
    eval_env = get_vec_env("SynthAtari-v0", {"reward_scale": 1}, N)
    kwp = dict(qf=qf, start_epsilon=1, end_epsilon=0.1, decay_frames=1000000, action_shape=A)
    pf = policies.EpsilonGreedyQRDQNDiscretePolicy(quantile_num=Q, **kwp) if Q > 1 else policies.EpsilonGreedyDQNDiscretePolicy(**kwp)
    buf = MemoryEfficientReplayBuffer(ROWS * N, env_nums=N, min_episode_frames=999) if dedup else BaseReplayBuffer(ROWS * N, env_nums=N)
    col = VecCollector(env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=dev, epoch_frames=N * STEPS,
                       max_episode_frames=999)
    kw = dict(qf=qf, pf=pf, qlr=2.5e-4, env=env, replay_buffer=buf, collector=col, logger=CountingLog(), discount=0.99,
              num_epochs=1, batch_size=B, device=dev, save_dir=None, tau=0.005, opt_times=OPT)
    agent = QRDQN(quantile_num=Q, **kw) if Q > 1 else DQN(**kw)
    for _ in range(4):                                                   # (one by one, eager epoch, captured epoch, replayed)
        col.rollout(STEPS); agent.update_per_epoch()
    log = agent.logger
    log.drain(); torch.cuda.synchronize()
    import gc
    gc.collect(); gc.freeze()       # (a generation-2 collection costs ~70 ms with torch loaded: keep it out of either loop)
    # (1) the two phases timed apart: a host wait after each (the update window includes reading its info dicts)
    tc = tu = 0.0
    for _ in range(epochs):
        a = time.perf_counter(); col.rollout(STEPS); torch.cuda.synchronize()
        b = time.perf_counter(); agent.update_per_epoch(); log.drain(); torch.cuda.synchronize()
        tc += b - a; tu += time.perf_counter() - b
    # (2) whole epochs in RLAlgo.train's order (rl_algo.py:111-118): collect, update, then look at the collector's result;
    # the update's info dicts are read when the logger would need them (here: one epoch later), nothing else waits
    log.updates = 0
    t0 = time.perf_counter(); seen = 0
    for _ in range(epochs):
        res = col.train_one_epoch()
        log.drain()
        agent.update_per_epoch()
        seen += len(res["train_rewards"])
    log.drain(); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    assert log.updates == epochs * OPT
    out = {"workload": "%s cfg5: %d envs, 84x84x4 u8 frames, %d-row replay, B=%d, conv 16/32/64 + fc512%s"
                       % ("QRDQN" if Q > 1 else "DQN", N, ROWS, B, ", Q=%d" % Q if Q > 1 else ""),
           "env_steps_per_s": epochs * N * STEPS / el,
           "env_steps_per_s_phases_timed_apart": epochs * N * STEPS / (tc + tu), "updates_per_s": epochs * OPT / tu,
           "ms_per_update": 1e3 * tu / (epochs * OPT), "ms_per_vector_step": 1e3 * tc / (epochs * STEPS),
           "update_gflop": 38e-3 * B, "replay": "frame-dedup" if dedup else "plain",
           "replay_frame_bytes": int(buf._stream.numel()) if dedup else int(buf._obs.numel() + buf._next_obs.numel())}
    if dedup:
        buf.check_overrun()
    if cpu:
        out["cpu_oracle_ms_per_update"] = cpu_baseline_via_bench("qrdqn" if Q > 1 else "dqn")["value"]
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--quantiles", type=int, default=1)
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--dedup", action="store_true", help="frame-deduplicating replay (MemoryEfficientReplayBuffer)")
    a = ap.parse_args()
    run(a.quantiles, a.epochs, a.cpu, a.dedup)
