"""Development aid: 10 launches of the dense-layer forward / input-gradient / weight-gradient GEMM at 4096^3 for a
rocprofv3 --pmc pass (TRL_GEMM_TILE=64 / 128 pins the workgroup tile)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd import _C
dev = torch.device("cuda:0")
torch.manual_seed(0)
M = K = N = 4096
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.02
y = _C.linear_fwd(x, w, None, _C.ACT_NONE)
dy = torch.randn(M, N, device=dev)
ws = torch.empty(_C.lib().trl_linear_bwd_weight_workspace(M, K, N), device=dev)
dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
for _ in range(6):
    _C.linear_fwd(x, w, None, _C.ACT_NONE)
    if "--all" in sys.argv:
        _C.linear_bwd_input(dy, y, 1, w)
        _C.linear_bwd_weight(dy, y, 1, x, dw=dw, db=db, workspace=ws)
torch.cuda.synchronize()
