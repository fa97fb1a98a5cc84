"""bench.py itself at N > 1 on ONE GPU (BASELINE cfg 4's launch path): plain `python bench.py --gpus 2`
with no launcher environment must spawn its own ranks, rendezvous, run the env-sharded iteration and print ONE valid JSON
line.  Both ranks are pinned to cuda:0 through the test-only TRL_BENCH_DEVICE_MAP; RCCL refuses two ranks per device, so
the process group is gloo and the two routes below are what a single GPU can exercise: the library's peer transport
(hipIpc-mapped buffers, gradient SUM inside the fold / clip / Adam launch) and the torch.distributed fallback."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env, gpus=2, extra_args=()):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(TRL_BENCH_DEVICE_MAP=",".join(["0"] * gpus), **extra_env)
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "3",
                          "--no-cpu-baseline", "--no-secondary", *extra_args], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                                       # rank 0 prints ONE line, the other rank nothing
    return json.loads(lines[0]), res.stderr


def test_plain_python_two_ranks_over_the_peer_transport():
    out, err = _bench({})
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    cfg = out["config"]
    assert cfg["transport"] == "peer", (cfg, err[-2000:])
    assert cfg["peer_self_check"].startswith("passed") and cfg["guarded_iterations"].startswith("3 completed")
    assert cfg["process_group_backend"] == "gloo" and cfg["launcher"].startswith("bench.py spawned")
    assert out["value"] > 0 and abs(out["value"] - 2 * 2048 * 128 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]
    assert all(v > 0 for v in cfg["collective_us"].values())
    assert out["roofline"]["launches_timed"] > 0 and 0 < out["roofline"]["frac"] < 1
    # the reference's exploration-noise stream at world size 2: every rank draws its rows of each
    # step's (N_total, A) tensor one rollout ahead; timed like the headline, and the headline when it costs <= 1.15x
    assert out["parity_mode_ms_per_step"] > 0 and out["device_noise_ms_per_step"] > 0
    assert cfg["headline_mode"].startswith("parity") or cfg["headline_mode"].startswith("device")
    assert cfg["transport_requested"] == "auto"
    assert cfg["exchanges_per_iteration"]["c1_gradient_sum_44KB"] == 40


def test_plain_python_two_ranks_on_the_all_reduce_fallback():
    out, err = _bench({}, extra_args=("--transport", "rccl"))               # (no RCCL with two ranks per device: gloo carries it)
    cfg = out["config"]
    assert out["n_gpus"] == 2
    assert cfg["transport"] == "torch.distributed:gloo", (cfg, err[-2000:])
    assert cfg["peer_self_check"].startswith("disabled") and cfg["transport_requested"] == "rccl"
    assert all(v > 0 for v in cfg["collective_us"].values())


def test_cfg4_layout_eight_ranks_on_one_gpu():
    """BASELINE cfg 4 in its own shape as far as one GPU allows: `python bench.py --gpus 8`,
    8 self-spawned ranks x 2048 envs = 16 384 envs, every rank drawing its rows of the reference's (16384, 6) noise tensors,
    all eight hipIpc buffers mapped and self-checked (8 slots each) -- and since round 6 the iterations themselves on the
    PEER transport: the gradient SUM over eight slots inside every rank's fold / clip / Adam launch, graph-replayed, at
    cfg 4's real sizes.  Eight ranks share the device, so each waiting fold launch is bounded to 16 blocks
    (`config.wait_footprint`; eight GPUs run the same kernel with one block per 64 parameters).  The figure itself is a
    correctness artefact: eight processes time-slice one GPU."""
    out, err = _bench({}, gpus=8)
    cfg = out["config"]
    assert out["n_gpus"] == 8 and out["steps"] == 3 and out["scaling"] == "weak"
    assert abs(out["value"] - 8 * 2048 * 128 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]
    assert cfg["peer_self_check"].startswith("passed on every rank"), (cfg, err[-2000:])
    assert cfg["ranks_per_device"] == 8 and "peer_transport_not_used" not in cfg
    assert cfg["transport"] == "peer" and cfg["transport_vote"] == "peer", (cfg, err[-2000:])
    assert cfg["guarded_iterations"].startswith("3 completed") and cfg["wait_footprint"].startswith("16 blocks")
    assert cfg["peer_buffer"].startswith("uncached") or cfg["peer_buffer"].startswith("plain")
    assert cfg["launcher"].startswith("bench.py spawned")
    assert all(v > 0 for v in cfg["collective_us"].values())
    assert out["parity_mode_ms_per_step"] > 0 and out["device_noise_ms_per_step"] > 0      # world-8 reference noise was timed
    assert "CPUs per rank" in cfg["cpu_affinity"]
    assert cfg["exchanges_per_iteration"]["c1_gradient_sum_44KB"] == 40


def test_a_rank_that_dies_takes_the_job_down_with_its_exit_code():
    """Three ranks mapped onto two devices of which only one exists: rank 2 exits before the rendezvous completes; the
    launcher must stop the others and return non-zero instead of waiting for a collective forever."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(TRL_BENCH_DEVICE_MAP="0,0,63")
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "3", "--steps", "1", "--warmup", "1",
                          "--no-cpu-baseline", "--no-secondary"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=300)
    assert res.returncode != 0
    assert "wants cuda:63" in res.stderr
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.parametrize("transport,want", [("rccl", "rccl"), ("auto", "peer")])
def test_full_size_one_rank_with_the_collectives_forced_on(transport, want):
    """The cfg 2-sized iteration with every cross-rank exchange switched on at world size 1 (TRL_FORCE_COLLECTIVES=1, nccl
    process group, launcher-style environment): `--transport rccl` = fold -> ncclAllReduce through the C ABI -> clip + Adam,
    graph-captured after the child-process probe; `auto` = the peer buffer mapped onto itself inside the fold launch.  What
    the first real multi-GPU run executes per rank, at full size, on the one GPU the tests have."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               TRL_FORCE_COLLECTIVES="1")
    env.pop("TRL_BENCH_DEVICE_MAP", None)
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "3",
                          "--no-cpu-baseline", "--no-secondary", "--transport", transport], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    cfg = out["config"]
    assert cfg["transport"] == want, (cfg, res.stderr[-2000:])
    assert cfg["process_group_backend"] == "nccl" and cfg["transport_requested"] == transport
    assert out["n_gpus"] == 1 and out["value"] > 2e7                     # (a wedged or host-bound route would be far below)
    assert all(v > 0 for v in cfg["collective_us"].values())
