"""Time the rollout kernels alone at the benchmark shape; with a -DTRL_EXP_CLK build of the library
(TRL_LIB=<path>) also print the per-phase cycle totals of workgroup 0.  Development aid."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from torchrl_amd import _C  # noqa: E402

if os.environ.get("TRL_LIB"):
    _C.LIB_PATH = os.environ["TRL_LIB"]
import bench  # noqa: E402

PHASES = ["noise", "L1+tanh+stT", "barrier1", "L2+tanh+head", "barrier2", "mean+act+logp", "env+tanh", "book+store"]


def main():
    dev = torch.device("cuda:0")
    agent, col = bench.build_agent(dev, 1, 0)
    col.env.reset()
    T = col.sample_epoch_frames
    for _ in range(3):
        col.rollout(T)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); col.rollout(T); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    print("%s rollout (2 kernels) T=%d: mean %.1f us min %.1f us" % (os.environ.get("TRL_LIB", "default"), T, sum(ts) / len(ts), min(ts)))
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); col._launch(col.env, T, False, False, None); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    print("   no ring stores / no value pass: mean %.1f us min %.1f us" % (sum(ts) / len(ts), min(ts)))
    if "clk" in os.environ.get("TRL_LIB", ""):
        log = col._ep_log.cpu().reshape(-1)
        base = (col.EP_LOG_CAP - 16) * 3
        for mo in range(4):
            v = log[base + mo * 8: base + mo * 8 + 8] / T
            print("wave %d cycles/step: " % mo + "  ".join("%s %.0f" % (p, x) for p, x in zip(PHASES, v.tolist())) + "  | total %.0f" % v.sum().item())


if __name__ == "__main__":
    main()
