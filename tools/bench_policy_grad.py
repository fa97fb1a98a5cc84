"""Development aid: trl_sac_policy_grad_f32 at cfg 3's shape -- time per launch, and with a -DTRL_EXP_CLK build
(python torchrl_amd/build.py --exp clk -DTRL_EXP_CLK; TRL_LIB=torchrl_amd/lib/libtrl_hip_clk.so) the phase stamps."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd import _C

dev = torch.device("cuda:0")
B, H, D, A = 4096, 256, 17, 6
dys = [torch.randn(B, H, device=dev) for _ in range(2)]
ys = [torch.randn(B, H, device=dev).clamp_min(0) for _ in range(2)]
ws = [torch.randn(H, D + A, device=dev) * 0.1 for _ in range(2)]
head, eps = torch.randn(B, 2 * A, device=dev), torch.randn(B, A, device=dev)
act = torch.tanh(torch.randn(B, A, device=dev))
alpha = torch.tensor([0.2], device=dev)
big = torch.empty(64 << 20, device=dev)
run = lambda: _C.sac_policy_grad(head, eps, act, dys, ys, _C.ACT_RELU, ws, D, alpha, 1.0 / B, 1e-3, 1e-3, True)
for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for flush in (False, True):
    ts = []
    for _ in range(20):
        if flush:
            big.zero_()
        e0.record(); run(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print("us per launch (events, %s): min %.1f median %.1f" % ("caches flushed" if flush else "inputs warm", min(ts), sorted(ts)[10]))
lib = _C.lib()
if hasattr(lib, "trl_dbg_pg_clk"):
    out = (C.c_longlong * 16)()
    lib.trl_dbg_pg_clk(out)
    for g in range(2):
        t = [out[8 * g + k] for k in range(5)]
        print("wg %s: " % ("first" if g == 0 else "last") + "  ".join("%s %.2f" % (n, (t[k + 1] - t[k]) / 100.0) for k, n in
              enumerate(["issue row loads", "stage weights", "row dots", "butterfly + finish"])) + "  us")
