"""Host time of one block of the reference's exploration noise (cfg 2: 128 x 2048 x 6 float32 normals from the CPU torch
generator) drawn by P threads (torchrl_amd/collector/noise.py) against the single `torch.randn` call it reproduces."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd.collector import noise   # noqa: E402

n = 128 * 2048 * 6
out = torch.empty(n).pin_memory() if torch.cuda.is_available() else torch.empty(n)
res = {"elements": n, "cpu_count": os.cpu_count()}
for P in (1, 2, 4, 8, 16):
    ts = []
    for _ in range(8):
        t = time.perf_counter()
        noise.randn_into(out, threads=P)
        ts.append((time.perf_counter() - t) * 1e3)
    res["threads_%d_ms" % P] = round(min(ts), 3)
t = time.perf_counter()
noise.segment_states(torch.get_rng_state(), n, 8)
res["segment_states_ms"] = round((time.perf_counter() - t) * 1e3, 3)
torch.manual_seed(1)
want = torch.randn(n)
torch.manual_seed(1)
res["bit_exact"] = bool(torch.equal(noise.randn_into(torch.empty(n), threads=8), want))
print(json.dumps(res))
# env shards: this rank's rows of every step's (N_total, A) draw (rank 1 of `world`)
for world in (2, 8):
    blk = torch.empty(128, 2048, 6)
    for P in (1, 4, 8):
        ts = []
        for _ in range(6):
            t = time.perf_counter()
            noise.randn_shard_into(blk, 128, 2048, 2048 * world, 2048, 6, threads=P)
            ts.append((time.perf_counter() - t) * 1e3)
        res["world_%d_threads_%d_ms" % (world, P)] = round(min(ts), 3)
    t = time.perf_counter()
    noise.states_at(torch.get_rng_state(), [s * 2048 * world * 6 + 2048 * 6 for s in range(128)] + [128 * 2048 * world * 6])
    res["world_%d_states_pass_ms" % world] = round((time.perf_counter() - t) * 1e3, 3)
print(json.dumps(res))
