"""Worker of tests/test_dist_gpu.py::test_abi_collectives_*: the C-ABI collectives (include/trl_hip.h, trl_comm_*) on their
own.  argv: rank world port out_path backend.  world 2 -> two processes sharing cuda:0, gloo rendezvous, peer transport
only; world 1 -> nccl rendezvous, RCCL communicator + peer buffer mapped onto itself."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    rank, world, port, out, backend = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    import torch.distributed as td
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    if backend == "nccl":
        td.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        td.init_process_group("gloo", rank=rank, world_size=world)
    from torchrl_amd import _C, dist
    # (world > 2 on the one GPU of the tests: only the small stand-alone all-reduces run here, all ranks' launches fit together)
    assert dist.init_comm(dev, use_rccl=(backend == "nccl"), allow_shared_device=True), "peer transport did not come up"
    assert dist.peer_report()["ranks_per_device"] == world
    lib = _C.lib()
    assert bool(lib.trl_comm_has_rccl(dist.comm_handle())) == (backend == "nccl")
    res = {}
    for n in (1, 100, 4096, 4097, 11085, 12288, 20000):            # statistics region / gradient region / beyond
        x = (torch.arange(n, device=dev, dtype=torch.float32) % 13 - 6) * (rank + 1)
        for rep in range(3):                                         # both buffer halves, repeated epochs
            y = x.clone()
            dist.all_reduce_sum_(y)
        res["f32_%d" % n] = y.cpu().numpy()
    d = torch.zeros(40, 4, dtype=torch.float64, device=dev)
    d[:, 0], d[:, 1], d[:, 2], d[:, 3] = rank + 1.0, (rank + 1.0) ** 2, rank - 5.0, -(rank + 2.0)
    dist.reduce_adv_raw_(d)                                          # {sum, sumsq} SUM, {max, -min} MAX in one call
    res["adv_raw"] = d.cpu().numpy()
    info = torch.full((40, 24), float(rank + 1), dtype=torch.float64, device=dev)
    dist.reduce_info_(info)
    res["info"] = info.cpu().numpy()
    m = torch.tensor([float(rank) + 0.5], dtype=torch.float64, device=dev)
    dist.all_reduce_max_(m)
    res["max"] = m.cpu().numpy()
    # the collectives are plain launches: a captured sequence replays
    g = torch.cuda.CUDAGraph()
    z = torch.full((5000,), float(rank + 1), device=dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        dist.all_reduce_sum_(z.clone())                              # warm-up outside the capture
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    td.barrier()
    with torch.cuda.graph(g):
        dist.all_reduce_sum_(z)
    for _ in range(3):
        z.fill_(float(rank + 1))
        g.replay()
    torch.cuda.synchronize()
    res["graph"] = z[:4].cpu().numpy()
    dist.check_comm()
    np.savez(out, **res)
    td.barrier()
    dist.destroy_comm()
    td.destroy_process_group()


if __name__ == "__main__":
    main()
