"""Do independent branches of a captured HIP graph run concurrently on this runtime?  Two dense-layer GEMMs that each
leave part of the chip idle (FC1-shaped: 392 workgroups) -- back to back in one stream vs forked onto a side stream
inside the capture, replayed as a graph; plus the eager two-stream version."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchrl_amd import _C
dev = torch.device("cuda:0")
torch.manual_seed(0)
M, K, N = 512, 3136, 512
dy = torch.randn(M, N, device=dev); y = torch.randn(M, N, device=dev); x = torch.randn(M, K, device=dev)
w = torch.randn(N, K, device=dev) * 0.02
ws = torch.empty(_C.lib().trl_linear_bwd_weight_workspace(M, K, N), device=dev)
dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
side = torch.cuda.Stream(dev)


def a(): _C.linear_bwd_weight(dy, y, 1, x, dw=dw, db=db, workspace=ws)
def b(): return _C.linear_bwd_input(dy, y, 1, w)


def serial(n=8):
    for _ in range(n):
        a(); b()


def forked(n=8):
    cur = torch.cuda.current_stream(dev)
    for _ in range(n):
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            a()
        b()
        cur.wait_stream(side)


def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps / 8


serial(); forked(); torch.cuda.synchronize()
print("eager serial  : %.1f us per (weight-grad + input-grad) pair" % timed(serial))
print("eager forked  : %.1f us per pair" % timed(forked))
for name, fn in (("serial", serial), ("forked", forked)):
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream(dev)
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            fn()
    print("graph %s : %.1f us per pair" % (name, timed(g.replay)))
