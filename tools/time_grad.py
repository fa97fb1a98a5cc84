"""Time the dominant kernel (ppo_grad_kernel) alone at the benchmark shape, optionally with an
experimental build of the library (TRL_LIB=<path to .so>).  Development aid, not part of the product."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from torchrl_amd import _C  # noqa: E402

if os.environ.get("TRL_LIB"):
    _C.LIB_PATH = os.environ["TRL_LIB"]
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    agent, col = bench.build_agent(dev, 1, 0)
    col.env.reset()
    col.rollout(col.sample_epoch_frames)
    agent.current_epoch = 0
    agent.update_per_epoch()                     # fills advs / old_logp, warms everything
    eng = agent.engine()
    buf = agent.replay_buffer
    t = {"obs": buf._obs, "acts": buf._acts, "advs": buf._advs, "rets": buf._estimate_returns,
         "old_values": buf._values, "old_logp": buf._old_logp}
    eng._n_wg = lambda n: (eng.max_wg, _C.lib().trl_ppo_wg_split(17, 64, 6, (n + 15) // 16, eng.max_wg))                      # always the full grid, to expose the fixed cost
    for rows in (32, 16, 8, 2, 1):
        idx = np.random.permutation(128)[:4 * rows].reshape(4, rows).astype(np.int64)
        probes = []
        eng.probe = probes
        for _ in range(5):
            eng.run(t, idx, buf.env_nums)
        torch.cuda.synchronize()
        ms = np.array([s.elapsed_time(e) for s, e in probes][4:])
        print("%s grad kernel rows_mb=%2d (B=%6d): mean %.1f us  min %.1f us  (n=%d)" % (
            os.environ.get("TRL_LIB", "default"), rows, rows * buf.env_nums, ms.mean() * 1e3, ms.min() * 1e3, len(ms)))
        if "clk" in os.environ.get("TRL_LIB", ""):
            names = ["prologue", "L1+st", "L2+tanh", "head/loss/dW3/dH2", "dz2st+dH1+dz1st", "dW2", "dW1", "images", "fold"]
            part = eng.partial.cpu().numpy()
            half = part.shape[0] // 2
            for net, sl in (("pf", slice(0, half)), ("vf", slice(half, None))):
                c = part[sl, eng.p_stride - 16: eng.p_stride - 7].mean(axis=0)
                print("   %s wave-0 cycles: " % net + "  ".join("%s %.0f" % (n, x) for n, x in zip(names, c)) + "  | total %.0f" % c.sum())


if __name__ == "__main__":
    main()
