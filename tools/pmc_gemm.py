"""Development aid: the grouped SAC critic layer (6 x 4096 x 256 -> 256) forward / input-grad / weight-grad and the direct
first-conv kernels, 100 launches each, for a rocprofv3 --pmc pass (per-kernel MFMA busy share)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd import _C
dev = torch.device("cuda:0")
torch.manual_seed(0)
G, M, K, N = 6, 4096, 256, 256
xs = [torch.randn(M, K, device=dev) for _ in range(G)]
ws = [torch.randn(N, K, device=dev) * 0.05 for _ in range(G)]
bs = [torch.randn(N, device=dev) for _ in range(G)]
ys = _C.linear_fwd_group(xs, ws, bs, 1)
dys = [torch.randn(M, N, device=dev) for _ in range(G)]
dws = [torch.empty(N, K, device=dev) for _ in range(G)]; dbs = [torch.empty(N, device=dev) for _ in range(G)]
wsp = torch.empty(G * _C.lib().trl_linear_bwd_weight_workspace(M, K, N), device=dev)
frames = torch.randint(0, 256, (512, 4, 84, 84), dtype=torch.uint8, device=dev)
cw = torch.randn(16, 256, device=dev) * 0.05; cb = torch.randn(16, device=dev)
cy, _ = _C.conv_fwd_u8(frames, cw, cb, 8, 8, 4, 4, 1 / 255.0, -0.5, 1)
cdy = torch.randn_like(cy); cdw = torch.empty_like(cw); cdb = torch.empty_like(cb)
for _ in range(100):
    _C.linear_fwd_group(xs, ws, bs, 1)
    _C.linear_bwd_input_group(dys, ys, 1, ws)
    _C.linear_bwd_weight_group(dys, ys, 1, xs, dws, dbs, workspace=wsp)
    _C.conv_fwd_u8(frames, cw, cb, 8, 8, 4, 4, 1 / 255.0, -0.5, 1)
    _C.conv_bwd_weight_u8(cdy, cy, 1, frames, 8, 8, 4, 4, 1 / 255.0, -0.5, cdw, cdb)
torch.cuda.synchronize()
