"""Env-sharded PPO on the GPU: two ranks x 32 envs reproduce one process x 64 envs.

Both ranks run on cuda:0 and talk through gloo -- RCCL refuses two ranks on one device, and the round-end 8-GPU run is
not available to the tests -- so this covers everything of the multi-GPU path except the transport: env index blocks
with global seeds, the Philox noise keyed by the global env index, identical host index streams, C2 (advantage
statistics), C1 (gradient SUM before clipping, unfused reduce -> all-reduce -> clip + Adam) and C3 (logged statistics)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def _run(world, tmp_path):
    port = _free_port()
    outs = [str(tmp_path / ("w%d_r%d.npz" % (world, r))) for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_gpu_worker.py"), str(r), str(world), port, outs[r]],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    for p in procs:
        log, _ = p.communicate(timeout=600)
        assert p.returncode == 0, log[-3000:]
    return [np.load(o) for o in outs]


def test_two_ranks_reproduce_one_process(tmp_path):
    (single,) = _run(1, tmp_path)
    r0, r1 = _run(2, tmp_path)
    # the data path needs no collective: the ranks' rollout shards are the column blocks of the single-process rollout
    np.testing.assert_allclose(np.concatenate([r0["obs"], r1["obs"]], axis=1), single["obs"], atol=1e-6)
    np.testing.assert_allclose(np.concatenate([r0["rewards"], r1["rewards"]], axis=1), single["rewards"], atol=1e-6)
    # replicated parameters stay identical across ranks, and equal to the single-process run up to summation order
    assert np.array_equal(r0["pf"], r1["pf"]) and np.array_equal(r0["vf"], r1["vf"])
    np.testing.assert_allclose(r0["pf"], single["pf"], atol=2e-6)
    np.testing.assert_allclose(r0["vf"], single["vf"], atol=2e-6)
    # the logged statistics are the GLOBAL ones on every rank
    assert list(r0["keys"]) == list(single["keys"])
    np.testing.assert_allclose(r0["infos"], r1["infos"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(r0["infos"], single["infos"], rtol=2e-4, atol=2e-5)
