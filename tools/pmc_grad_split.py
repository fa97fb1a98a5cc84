"""The gradient kernel at B = rows x 2048 for a chosen number of time rows per minibatch, a few launches, nothing else:
run under `rocprofv3 --pmc ...` once with --rows 32 (the benchmark's 65 536 samples) and once with --rows 16, and the
difference of the two per-launch counter means is what the TILE LOOP costs for 2 048 tiles -- the remainder is the fixed
part of a launch (weights in, gradient images out, fold).  Used to split SQ_WAIT_ANY into in-loop and prologue / epilogue
shares (profiles/r03_grad_wait_split.txt).  Development aid, not part of the product."""
import argparse
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=32)
    ap.add_argument("--launches", type=int, default=24)
    a = ap.parse_args()
    import bench
    dev = torch.device("cuda:0")
    agent, col = bench.build_agent(dev, 1, 0)
    col.env.reset()
    col.rollout(col.sample_epoch_frames)
    agent.current_epoch = 0
    np.random.seed(0)
    agent.update_per_epoch()
    agent.logger.drain()
    eng = agent.engine()
    buf = agent.replay_buffer
    t = {"obs": buf._obs, "acts": buf._acts, "advs": buf._advs, "rets": buf._estimate_returns,
         "old_values": buf._values, "old_logp": buf._old_logp}
    k = 128 // a.rows
    idx = np.random.RandomState(1).permutation(128)[:k * a.rows].reshape(k, a.rows).astype(np.int64)
    probes = []
    eng.probe = probes                                  # eager launches (no graph)
    for _ in range(max(1, a.launches // k)):
        eng.run(t, idx, buf.env_nums)
    torch.cuda.synchronize()
    us = np.array([s.elapsed_time(e) for s, e in probes]) * 1e3
    print("rows %d: %d launches, mean %.2f us, min %.2f us" % (a.rows, len(us), us.mean(), us.min()))


if __name__ == "__main__":
    main()
