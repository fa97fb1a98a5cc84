import torch
x = torch.zeros(64, device="cuda")
def body(n=80):
    for _ in range(n): x.add_(1.0)
for _ in range(3): body()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); body(800); e.record(); torch.cuda.synchronize()
print("eager dependent tiny kernels: %.2f us each" % (s.elapsed_time(e) * 1e3 / 800))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body(800)
g.replay(); torch.cuda.synchronize()
s.record(); g.replay(); e.record(); torch.cuda.synchronize()
print("graph  dependent tiny kernels: %.2f us each" % (s.elapsed_time(e) * 1e3 / 800))
