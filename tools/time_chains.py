"""Time the 40 updates of an iteration alone (no rollout): the joint launch sequence against the two chains
(TRL_PPO_CHAINS), graph-replayed, wall clock around a device wait.  Development aid."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    agent, col = bench.build_agent(dev, 1, 0)
    col.env.reset()
    col.rollout(col.sample_epoch_frames)
    agent.current_epoch = 0
    np.random.seed(0)
    agent.update_per_epoch()
    eng = agent.engine()
    buf = agent.replay_buffer
    t = {"obs": buf._obs, "acts": buf._acts, "advs": buf._advs, "rets": buf._estimate_returns,
         "old_values": buf._values, "old_logp": buf._old_logp}
    idx = np.stack([np.random.RandomState(e).permutation(128).reshape(4, 32) for e in range(10)]).reshape(40, 32).astype(np.int64)
    for _ in range(4):
        eng.run(t, idx, buf.env_nums)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run(t, idx, buf.env_nums, defer=True)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    if os.environ.get("TRL_LIB", "").endswith("chainclk.so"):        # -DTRL_CHAIN_CLK build: the LAST update's workgroup stamps
        def stamps(part, scal, n):
            pad = part[:n, -3:].double().cpu().numpy()
            start = pad[:, 0] * float(1 << 40) + pad[:, 1] * float(1 << 20) + pad[:, 2]
            end = scal[:n, 7].cpu().numpy()
            return start / 100.0, end / 100.0
        n_wg, n_pf = eng._n_wg(32 * buf.env_nums)
        if eng.two_chains:
            sp, ep = stamps(eng.partial, eng.scal, n_pf)
            sv, ev = stamps(*eng._chain_rows, n_wg - n_pf)
        else:
            s_all, e_all = stamps(eng.partial, eng.scal, n_wg)
            sp, ep, sv, ev = s_all[:n_pf], e_all[:n_pf], s_all[n_pf:], e_all[n_pf:]
        t0 = min(sp.min(), sv.min())
        print("   last update: policy kernel %.1f .. %.1f us (wg passes mean %.1f max %.1f), value kernel %.1f .. %.1f us (mean %.1f max %.1f)"
              % (sp.min() - t0, ep.max() - t0, (ep - sp).mean(), (ep - sp).max(), sv.min() - t0, ev.max() - t0, (ev - sv).mean(), (ev - sv).max()), flush=True)
    print("chains=%s: 40 updates %.1f us (min %.1f) = %.2f us per update" % (
        os.environ.get("TRL_PPO_CHAINS", "two"), 1e6 * np.median(ts), 1e6 * min(ts), 1e6 * np.median(ts) / 40), flush=True)


if __name__ == "__main__":
    main()
