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
    for rows in (32,):
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
            names = {0: "prologue", 9: "top+prefetch", 10: "L1 mfma", 1: "L1 act+st", 11: "L2 mfma", 12: "L2 act", 2: "fetch next",
                     14: "H2st+head mfma", 18: "logp terms", 19: "shfl", 20: "ratio/loss", 21: "dout+DOS st", 15: "stats", 16: "dz2+dW3 mfma", 3: "act'+dz2 st", 17: "dH1 mfma", 4: "act'+dz1 st",
                     5: "dW2", 6: "dW1", 7: "images", 8: "fold"}
            part = eng.partial.cpu().numpy()
            n_pf = _C.lib().trl_ppo_wg_split(17, 64, 6, (rows * buf.env_nums + 15) // 16, eng.max_wg)
            for net, sl in (("pf", slice(0, n_pf)), ("vf", slice(n_pf, None))):
                c = part[sl, eng.p_stride - 40: eng.p_stride - 16].mean(axis=0)
                print("   %s wave-0 ticks: " % net + "  ".join("%s %.0f" % (names[k], c[k]) for k in names) + "  | total %.0f" % c.sum())


if __name__ == "__main__":
    main()
