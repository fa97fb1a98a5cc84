"""Development aid: the one-launch DQN head (trl_dqn_head_f32) at cfg 5's shape -- time per launch, and with a
-DTRL_EXP_CLK build (python torchrl_amd/build.py --exp clk -DTRL_EXP_CLK; TRL_LIB=torchrl_amd/lib/libtrl_hip_clk.so)
the 100 MHz phase stamps of the first and the last workgroup."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd import _C

dev = torch.device("cuda:0")
B, H, A = 512, 512, 6
h, hn = torch.randn(B, H, device=dev), torch.randn(B, H, device=dev)
w, wt = torch.randn(A, H, device=dev) * 0.1, torch.randn(A, H, device=dev) * 0.1
b, bt = torch.randn(A, device=dev), torch.randn(A, device=dev)
acts = torch.randint(0, A, (B,), device=dev).float()
rew, term = torch.randn(B, device=dev), torch.zeros(B, device=dev)
dw, db, sums = torch.zeros_like(w), torch.zeros_like(b), torch.zeros(3, dtype=torch.float64, device=dev)
ws = _C.dqn_head_workspace(H, A, dev)
big = torch.empty(64 << 20, device=dev)
for _ in range(5):
    _C.dqn_head(h, hn, w, b, wt, bt, acts, rew, term, 0.99, dw, db, sums, ws)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(20):
    big.zero_()                                    # dirty lines in every L2, as inside an update
    e0.record()
    _C.dqn_head(h, hn, w, b, wt, bt, acts, rew, term, 0.99, dw, db, sums, ws)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
print("us per launch (events): min %.1f median %.1f" % (min(ts), sorted(ts)[len(ts) // 2]))
lib = _C.lib()
if hasattr(lib, "trl_dbg_dqh_clk"):
    out = (C.c_longlong * 16)()
    lib.trl_dbg_dqh_clk(out)
    for g in range(2):
        t = [out[8 * g + k] for k in range(7)]
        print("wg %2d: " % (0 if g == 0 else 63) + "  ".join("%s %.2f" % (n, (t[k + 1] - t[k]) / 100.0) for k, n in
              enumerate(["stage", "passes", "partial out", "bias + sums out", "fold", "sums"])) + "  us")
