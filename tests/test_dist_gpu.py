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


def _run(world, tmp_path, worker="_dist_gpu_worker.py", extra=(), env=None, _attempt=0):
    port = _free_port()
    outs = [str(tmp_path / ("%s_w%d_r%d%s.npz" % (worker[:-3], world, r, "_".join(extra)))) for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, worker), str(r), str(world), port, outs[r], *extra],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                              env=None if env is None else dict(os.environ, **env)) for r in range(world)]
    logs = [p.communicate(timeout=600)[0] for p in procs]
    if _attempt == 0 and any(p.returncode != 0 for p in procs) and any("EADDRINUSE" in l or "address already in use" in l for l in logs):
        return _run(world, tmp_path, worker, extra, env, _attempt=1)   # (the port was free when picked, not reserved: once more)
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    return [dict(np.load(o)) for o in outs]                           # (materialised: a later run with the same arguments rewrites the files)


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


def test_eight_ranks_reproduce_one_process(tmp_path):
    """BASELINE cfg 4's partition at test size: 8 ranks x 8 envs against one process x 64 envs, all on cuda:0 over gloo
    (the host-staged all-reduce route; the same eight ranks over the peer transport:
    test_eight_ranks_over_the_peer_transport_replay_a_graph).
    Device noise keyed by the global env index, then the reference's CPU noise stream with every rank drawing only its
    rows of each step's (64, 6) tensor one rollout ahead (world-8 offsets through the real collector)."""
    (single,) = _run(1, tmp_path)
    ranks = _run(8, tmp_path)
    np.testing.assert_allclose(np.concatenate([r["obs"] for r in ranks], axis=1), single["obs"], atol=1e-6)
    np.testing.assert_allclose(np.concatenate([r["rewards"] for r in ranks], axis=1), single["rewards"], atol=1e-6)
    for r in ranks[1:]:
        assert np.array_equal(r["pf"], ranks[0]["pf"]) and np.array_equal(r["vf"], ranks[0]["vf"])
        np.testing.assert_allclose(r["infos"], ranks[0]["infos"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(ranks[0]["pf"], single["pf"], atol=2e-6)
    np.testing.assert_allclose(ranks[0]["vf"], single["vf"], atol=2e-6)
    np.testing.assert_allclose(ranks[0]["infos"], single["infos"], rtol=2e-4, atol=2e-5)
    # the reference's noise stream at world 8
    (single_h,) = _run(1, tmp_path, extra=("plain", "host_perstep"))
    ranks_h = _run(8, tmp_path, extra=("plain", "host_prefetch"))
    acts = np.concatenate([r["acts"] for r in ranks_h], axis=2)             # (epoch, T, N_total, A)
    assert np.array_equal(acts[0], single_h["acts"][0])                      # same parameters: the same actions, bit for bit
    np.testing.assert_allclose(acts, single_h["acts"], atol=2e-5)
    for r in ranks_h:
        assert np.array_equal(r["tail"], single_h["tail"])                   # the CPU stream ends where the draws for ALL envs leave it
        assert np.array_equal(r["pf"], ranks_h[0]["pf"]) and int(r["prefetched"]) >= 1
    np.testing.assert_allclose(ranks_h[0]["pf"], single_h["pf"], atol=2e-6)


def test_eight_ranks_at_cfg4_size_reproduce_one_process(tmp_path):
    """The same at BASELINE cfg 4's real sizes: 16 384 envs x 128 steps, 8 ranks x 2048 envs with minibatches of 32 time rows
    (65 536 samples per rank, 524 288 globally) against ONE process holding all 16 384 envs.  Every rank's rollout buffers
    are the column blocks of the one-process buffers, parameters are bit-identical across the eight ranks after two
    epochs of 8 updates and follow the one-process parameters up to the order of the gradient sum."""
    sizes = {"TRL_TEST_SIZES": "16384,128,32,2"}
    (single,) = _run(1, tmp_path, env=sizes)
    ranks = _run(8, tmp_path, env=sizes)
    assert single["obs"].shape == (128, 16384, 17) and ranks[0]["obs"].shape == (128, 2048, 17)
    np.testing.assert_allclose(np.concatenate([r["obs"] for r in ranks], axis=1), single["obs"], atol=2e-6)
    np.testing.assert_allclose(np.concatenate([r["rewards"] for r in ranks], axis=1), single["rewards"], atol=2e-6)
    # the first epoch's actions come from identical parameters: the shards are the column blocks, bit for bit
    assert np.array_equal(np.concatenate([r["acts"][0] for r in ranks], axis=1), single["acts"][0])
    for r in ranks[1:]:
        assert np.array_equal(r["pf"], ranks[0]["pf"]) and np.array_equal(r["vf"], ranks[0]["vf"])
        np.testing.assert_allclose(r["infos"], ranks[0]["infos"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(ranks[0]["pf"], single["pf"], atol=2e-6)
    np.testing.assert_allclose(ranks[0]["vf"], single["vf"], atol=2e-6)
    np.testing.assert_allclose(ranks[0]["infos"], single["infos"], rtol=2e-4, atol=2e-5)


def test_two_ranks_reproduce_one_process_a2c(tmp_path):
    """A2C logs value-prediction statistics (a2c.py:86-105): their sums / extrema are pooled over the ranks as well."""
    (single,) = _run(1, tmp_path, extra=("a2c",))
    r0, r1 = _run(2, tmp_path, extra=("a2c",))
    assert np.array_equal(r0["pf"], r1["pf"]) and np.array_equal(r0["vf"], r1["vf"])
    np.testing.assert_allclose(r0["pf"], single["pf"], atol=2e-6)
    np.testing.assert_allclose(r0["vf"], single["vf"], atol=2e-6)
    assert "v_pred/max" in list(r0["keys"])
    np.testing.assert_allclose(r0["infos"], r1["infos"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(r0["infos"], single["infos"], rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("algo,p_tol,i_tol", [("vmpo", 2e-5, 2e-3), ("trpo", 5e-3, None)])
def test_two_ranks_gather_the_minibatch_for_global_steps(tmp_path, algo, p_tol, i_tol):
    """V-MPO's top half by advantage and TRPO's natural-gradient solve are steps over the GLOBAL minibatch: the ranks
    gather their env shards and run the update replicated, so parameters are bit-identical across ranks and follow the
    single-process run (TRPO up to the conditioning of its ten-iteration fp32 CG, see test_trpo_gpu.py)."""
    (single,) = _run(1, tmp_path, extra=(algo,))
    r0, r1 = _run(2, tmp_path, extra=(algo,))
    assert np.array_equal(r0["pf"], r1["pf"]) and np.array_equal(r0["vf"], r1["vf"])
    assert np.array_equal(r0["infos"], r1["infos"])
    np.testing.assert_allclose(r0["pf"], single["pf"], atol=p_tol)
    np.testing.assert_allclose(r0["vf"], single["vf"], atol=p_tol)
    if i_tol is not None:
        np.testing.assert_allclose(r0["infos"], single["infos"], rtol=i_tol, atol=i_tol)


@pytest.mark.parametrize("mode", ["", "peer"])
def test_two_ranks_draw_their_rows_of_the_reference_noise_stream(tmp_path, mode):
    """The reference's exploration noise at world size 2: every vector step the reference draws ONE
    (N_total, A) tensor from the CPU generator (torchrl/policies/distribution.py:60-76); each rank produces only ITS rows of
    it from the engine state at their position in the stream (collector/noise.py), one rollout ahead when prefetching.
    The ranks' buffers must be the column blocks of the single-process run that makes the reference's literal per-step
    draws -- torch.equal, every epoch -- and the generator must end where the draws for all envs leave it."""
    extra = (mode or "plain",)
    (single,) = _run(1, tmp_path, extra=("plain", "host_perstep"))
    (single_block,) = _run(1, tmp_path, extra=("plain", "host"))
    assert np.array_equal(single_block["acts"], single["acts"]) and np.array_equal(single_block["tail"], single["tail"])
    for opt in ("host", "host_prefetch"):
        r0, r1 = _run(2, tmp_path, extra=extra + (opt,))
        acts = np.concatenate([r0["acts"], r1["acts"]], axis=2)      # (epoch, T, N_total, A)
        assert np.array_equal(acts[0], single["acts"][0]), opt         # same parameters: the same actions, bit for bit
        np.testing.assert_allclose(acts, single["acts"], atol=2e-5)    # later epochs: parameters differ by summation order
        np.testing.assert_allclose(np.concatenate([r0["obs"], r1["obs"]], axis=1), single["obs"], atol=2e-5)
        assert np.array_equal(r0["tail"], single["tail"]) and np.array_equal(r1["tail"], single["tail"]), opt
        assert np.array_equal(r0["pf"], r1["pf"]) and np.array_equal(r0["vf"], r1["vf"])
        np.testing.assert_allclose(r0["pf"], single["pf"], atol=2e-6)
        if opt == "host_prefetch":                                    # blocks did arrive through the prefetcher
            assert int(r0["prefetched"]) >= 1 and int(r1["prefetched"]) >= 1, (int(r0["prefetched"]), int(r1["prefetched"]))


def test_rccl_sequence_replays_as_a_graph(tmp_path):
    """TRL_GRAPH_COLLECTIVES=1: the multi-rank launch sequence (partial fold -> RCCL all-reduce -> clip + Adam, step
    count and learning rates on the device) captured into a HIP graph and replayed, on a one-rank nccl group with the
    collectives forced on; it must reproduce the plain single-process run."""
    (single,) = _run(1, tmp_path)
    (forced,) = _run(1, tmp_path, extra=("nccl_graph",), env={"TRL_FORCE_COLLECTIVES": "1", "TRL_GRAPH_COLLECTIVES": "1"})
    np.testing.assert_allclose(forced["obs"], single["obs"], atol=1e-6)
    np.testing.assert_allclose(forced["pf"], single["pf"], atol=2e-6)
    np.testing.assert_allclose(forced["vf"], single["vf"], atol=2e-6)
    np.testing.assert_allclose(forced["infos"], single["infos"], rtol=2e-4, atol=2e-5)


def test_two_ranks_over_the_peer_transport_replay_a_graph(tmp_path):
    """The library's own communicator (trl_comm_*, include/trl_hip.h) between two processes sharing cuda:0: peer-mapped
    hipIpc buffers, the gradient SUM inside the fold / clip / Adam launch (trl_ppo_reduce_adam_xrank_f32), the
    statistics through the one-kernel all-reduce -- and the whole sequence graph-replayed from the third epoch on.  Same
    bar as the gloo run: bit-identical parameters across ranks, the single-process run up to summation order."""
    (single,) = _run(1, tmp_path)
    r0, r1 = _run(2, tmp_path, extra=("peer",))
    assert int(r0["peer"]) == 1 and int(r1["peer"]) == 1 and int(r0["graph"]) == 1
    np.testing.assert_allclose(np.concatenate([r0["obs"], r1["obs"]], axis=1), single["obs"], atol=1e-6)
    assert np.array_equal(r0["pf"], r1["pf"]) and np.array_equal(r0["vf"], r1["vf"])
    np.testing.assert_allclose(r0["pf"], single["pf"], atol=2e-6)
    np.testing.assert_allclose(r0["vf"], single["vf"], atol=2e-6)
    np.testing.assert_allclose(r0["infos"], r1["infos"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(r0["infos"], single["infos"], rtol=2e-4, atol=2e-5)


def test_two_update_chains_across_ranks_equal_the_joint_exchange(tmp_path):
    """Env shards on two ranks, peer transport: the critic's and the actor's updates as two launch sequences per rank --
    each fold launch carrying its own network's gradient SUM over ranks (trl_ppo_reduce_adam_xrank_net_f32: the policy's
    exchanges counted in ctl[4], the value function's in ctl[6], each network its own granules) -- against the joint
    sequence (TRL_PPO_CHAINS_XRANK=0: one exchange for both networks per update).  Same folds, same rank order of the
    sums, same Adam arithmetic: every buffer, parameter and logged statistic bit for bit, on both ranks."""
    a0, a1 = _run(2, tmp_path, extra=("peer",), env={"TRL_PPO_CHAINS_XRANK": "1"})
    b0, b1 = _run(2, tmp_path, extra=("peer",))                        # (ranks sharing a device: the joint exchange is the default)
    assert int(a0["chains"]) == 1 and int(b0["chains"]) == 0 and int(b0["graph"]) == 1
    for k in ("pf", "vf", "infos", "obs"):
        assert np.array_equal(a0[k], b0[k]) and np.array_equal(a1[k], b1[k]), k


def test_two_update_chains_across_ranks_at_full_size(tmp_path):
    """The same at the headline's per-rank sizes -- two ranks x 2048 envs x 128 steps, minibatches of 65 536 samples per
    rank, i.e. full-chip gradient grids split 152 + 104, four epochs (eager, two captures, a replay) -- where the next
    rollout really runs beside the value chain: it writes the shadow `obs` tensor, its value pass waits for the value
    chain's end event behind the cross-rank reduction of the statistics.  Rollout buffers of the last epoch (the rewards
    carry the bootstrap values of the value function), parameters and statistics: bit for bit those of the joint exchange."""
    sizes = {"TRL_TEST_SIZES": "4096,128,32,4"}
    # (every value chain held back by ~2 ms of device spin, as in test_two_update_chains_match_the_joint_sequence: the rollout
    # behind the policy chain is through before the value chain has read the previous rollout's observations)
    a0, a1 = _run(2, tmp_path, extra=("peer",), env=dict(sizes, TRL_PPO_CHAINS_XRANK="1", TRL_TEST_VALUE_CHAIN_DELAY="5000000"))
    b0, b1 = _run(2, tmp_path, extra=("peer",), env=dict(sizes, TRL_PPO_CHAINS_XRANK="0"))
    assert a0["obs"].shape == (128, 2048, 17) and int(a0["chains"]) == 1 and int(b0["chains"]) == 0
    for k in ("pf", "vf", "infos", "obs", "rewards", "acts"):
        assert np.array_equal(a0[k], b0[k]) and np.array_equal(a1[k], b1[k]), k


def test_a_missing_rank_trips_the_flag_and_the_check_does_not_idle_the_device(tmp_path):
    """A peer wait that nobody answers gives up after TRL_COMM_TIMEOUT_S instead of hanging the GPU, records whose granules
    were missing, and `dist.check_comm(peek=True)` -- the once-per-iteration check of the update loop, a 4-byte read on a
    stream of the communicator's own (trl_comm_error_peek) -- raises with that detail and clears the flag."""
    r0, _ = _run(2, tmp_path, extra=("peer_timeout",), env={"TRL_COMM_TIMEOUT_S": "1"})
    assert int(r0["raised"]) == 1 and int(r0["again"]) == 0, r0
    assert "rank 0 waited for rank 1's" in str(r0["text"]) and "granules of exchange" in str(r0["text"]), str(r0["text"])


@pytest.mark.parametrize("chains", ["joint", "two"])
def test_eight_ranks_over_the_peer_transport_replay_a_graph(tmp_path, chains):
    """BASELINE cfg 4's exchange at test size, for real: 8 processes x 8 envs sharing cuda:0, eight
    hipIpc-mapped buffers of 8 slots x 2 halves, the gradient SUM over the eight slots INSIDE the fold / clip / Adam launch
    (trl_ppo_reduce_adam_xrank_f32: push granules to seven peers, poll eight slots, norm rendezvous, Adam), the statistics
    through the one-kernel all-reduce, the sequence graph-replayed from the third epoch on (both epoch-parity halves of the
    buffers are used).  The ranks share one device, so every rank's waiting fold launch is bounded to CUs / 16 blocks
    (trl_comm_set_wait_footprint: a block then folds several 64-parameter jobs in turn) -- the same kernel that runs with
    one block per job on eight GPUs.  Same bar as two ranks: parameters bit-identical across the eight ranks, the
    single-process run up to the order of the gradient sum."""
    (single,) = _run(1, tmp_path)
    # `two`: every rank runs the critic's and the actor's updates as two launch sequences, i.e. sixteen sequences of waiting
    # fold launches on the one device, each network's exchange in its own granules (what one-rank-per-GPU runs do by default)
    ranks = _run(8, tmp_path, extra=("peer",), env={"TRL_PPO_CHAINS_XRANK": "1" if chains == "two" else "0"})
    for r in ranks:
        assert int(r["peer"]) == 1 and int(r["graph"]) == 1 and int(r["chains"]) == (chains == "two")
    np.testing.assert_allclose(np.concatenate([r["obs"] for r in ranks], axis=1), single["obs"], atol=1e-6)
    for r in ranks[1:]:
        assert np.array_equal(r["pf"], ranks[0]["pf"]) and np.array_equal(r["vf"], ranks[0]["vf"])
        np.testing.assert_allclose(r["infos"], ranks[0]["infos"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(ranks[0]["pf"], single["pf"], atol=2e-6)
    np.testing.assert_allclose(ranks[0]["vf"], single["vf"], atol=2e-6)
    np.testing.assert_allclose(ranks[0]["infos"], single["infos"], rtol=2e-4, atol=2e-5)


def test_fold_clip_adam_launch_does_not_depend_on_its_grid(tmp_path):
    """trl_comm_set_wait_footprint only changes how the 2 x 90 fold jobs are dealt to blocks: two ranks over the peer
    transport with the fold launch bounded to 3 blocks reproduce the run at the default footprint bit for bit."""
    a0, a1 = _run(2, tmp_path, extra=("peer",))
    b0, b1 = _run(2, tmp_path, extra=("peer",), env={"TRL_TEST_WAIT_BLOCKS": "3"})
    for k in ("pf", "vf", "infos", "obs"):
        assert np.array_equal(a0[k], b0[k]) and np.array_equal(a1[k], b1[k]), k


def test_one_rank_rccl_communicator_with_peer_transport(tmp_path):
    """World size 1 on the nccl backend with the collectives forced on: ncclCommInitRank through the C ABI, the peer
    buffer mapped onto itself, the cross-rank launch sequence graph-replayed; it must reproduce the plain run."""
    (single,) = _run(1, tmp_path)
    (forced,) = _run(1, tmp_path, extra=("nccl_peer",), env={"TRL_FORCE_COLLECTIVES": "1"})
    assert int(forced["peer"]) == 1 and int(forced["graph"]) == 1
    np.testing.assert_allclose(forced["pf"], single["pf"], atol=2e-6)
    np.testing.assert_allclose(forced["vf"], single["vf"], atol=2e-6)
    np.testing.assert_allclose(forced["infos"], single["infos"], rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("world,backend", [(2, "gloo"), (1, "nccl"), (8, "gloo")])
def test_abi_collectives_small_mid_large_and_graph_replay(tmp_path, world, backend):
    """trl_allreduce_sum_f32 / trl_allreduce_f64 through torchrl_amd.dist on their own: the statistics region (<= 4096 words),
    the gradient region (<= 12 288 floats), beyond it (RCCL on the nccl group, torch.distributed on the gloo one), the
    mixed SUM / MAX statistics rows, and a captured all-reduce replayed three times.  world = 8: BASELINE cfg 4's layout --
    eight processes, eight hipIpc-mapped buffers of 8 slots x 2 halves each, every granule summed over the 8 slots in rank
    order (all results exact: the test values are small integers)."""
    outs = _run(world, tmp_path, "_dist_gpu_worker_comm.py", (backend,))
    tot = sum(r + 1 for r in range(world))
    for r, o in enumerate(outs):
        for n in (1, 100, 4096, 4097, 11085, 12288, 20000):
            want = (np.arange(n, dtype=np.float32) % 13 - 6) * tot
            assert np.array_equal(o["f32_%d" % n], want), (r, n)
        raw = o["adv_raw"]
        assert np.all(raw[:, 0] == tot) and np.all(raw[:, 1] == sum((q + 1.0) ** 2 for q in range(world)))
        assert np.all(raw[:, 2] == world - 1 - 5.0) and np.all(raw[:, 3] == -2.0)
        info = o["info"]
        for c in range(24):
            assert np.all(info[:, c] == (tot if c in (0, 1, 2, 7, 12, 13) else world)), c
        assert o["max"][0] == world - 0.5
        assert np.all(o["graph"] == tot)


def test_two_ranks_share_the_observation_normaliser(tmp_path):
    """obs_norm with env shards: every step the ranks pool their batch moments (all-reduce) before the Chan merge, so
    both hold the statistics of ALL envs -- the single process keeps them inside the cooperative rollout kernel."""
    (single,) = _run(1, tmp_path, extra=("obs_norm",))
    r0, r1 = _run(2, tmp_path, extra=("obs_norm",))
    assert np.array_equal(r0["norm_state"], r1["norm_state"])
    np.testing.assert_allclose(r0["norm_state"], single["norm_state"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(np.concatenate([r0["obs"], r1["obs"]], axis=1), single["obs"], atol=5e-5)
    assert np.array_equal(r0["pf"], r1["pf"])
    np.testing.assert_allclose(r0["pf"], single["pf"], atol=1e-5)
    np.testing.assert_allclose(r0["infos"], single["infos"], rtol=2e-3, atol=2e-4)


@pytest.mark.parametrize("noise", ["device", "host"])
def test_two_ranks_reproduce_one_process_sac(tmp_path, noise):
    """Twin-Q SAC with the envs and the replay sharded over two ranks: exploration and update noise are the ranks' blocks
    of the draw for all envs, the temperature step sees the global batch, gradients are summed before clipping."""
    (single,) = _run(1, tmp_path, "_dist_gpu_worker_sac.py", (noise,))
    r0, r1 = _run(2, tmp_path, "_dist_gpu_worker_sac.py", (noise,))
    np.testing.assert_allclose(np.concatenate([r0["obs"], r1["obs"]], axis=1), single["obs"], atol=2e-5)
    np.testing.assert_allclose(np.concatenate([r0["acts"], r1["acts"]], axis=1), single["acts"], atol=2e-5)
    assert np.array_equal(r0["flat"], r1["flat"]) and np.array_equal(r0["tflat"], r1["tflat"])
    np.testing.assert_allclose(r0["flat"], single["flat"], atol=5e-6)
    np.testing.assert_allclose(r0["tflat"], single["tflat"], atol=5e-6)
    np.testing.assert_allclose(r0["log_alpha"], single["log_alpha"], atol=2e-6)
    assert list(r0["keys"]) == list(single["keys"])
    np.testing.assert_allclose(r0["infos"], r1["infos"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(r0["infos"], single["infos"], rtol=5e-4, atol=5e-5)


@pytest.mark.parametrize("algo", ["td3", "dqn"])
def test_two_ranks_reproduce_one_process_td3_dqn(tmp_path, algo):
    """TD3 (target-smoothing and exploration noise sharded like the envs, per-network gradient SUM) and DQN on uint8
    frames (epsilon-greedy host draws sharded, conv-net gradient SUM) with two ranks against one process."""
    (single,) = _run(1, tmp_path, "_dist_gpu_worker_sac.py", ("device", algo))
    r0, r1 = _run(2, tmp_path, "_dist_gpu_worker_sac.py", ("device", algo))
    np.testing.assert_allclose(np.concatenate([r0["acts"], r1["acts"]], axis=1), single["acts"], atol=2e-5)
    np.testing.assert_allclose(np.concatenate([r0["rewards"], r1["rewards"]], axis=1), single["rewards"], atol=1e-6)
    assert np.array_equal(r0["flat"], r1["flat"]) and np.array_equal(r0["tflat"], r1["tflat"])
    np.testing.assert_allclose(r0["flat"], single["flat"], atol=5e-6)
    np.testing.assert_allclose(r0["tflat"], single["tflat"], atol=5e-6)
    assert list(r0["keys"]) == list(single["keys"])
    np.testing.assert_allclose(r0["infos"], r1["infos"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(r0["infos"], single["infos"], rtol=5e-4, atol=5e-5)


def test_link_preflight_reports_every_pair_of_the_devices_present():
    """bench.py's pre-flight for `--gpus N`: peer access and link type for every ordered pair through
    trl_comm_link_info (hipDeviceCanAccessPeer + hipExtGetLinkTypeAndHopCount).  On the one-GPU test box: no pairs, the
    diagonal answers, an index past the device count is rejected."""
    import ctypes
    import torch
    from torchrl_amd import _C, dist
    n = torch.cuda.device_count()
    out = dist.link_preflight(range(n))
    assert out["pairs"] == n * (n - 1) and isinstance(out["peer_access_all"], bool)
    assert sum(out["links"].values()) + len(out["no_access"]) == out["pairs"]
    info = (ctypes.c_int32 * 3)()
    assert _C.lib().trl_comm_link_info(0, 0, info) == 0 and info[0] == 1
    assert _C.lib().trl_comm_link_info(0, n, info) != 0
    if n > 1:
        assert out["max_hops"] >= 1 and set(out["links"]) <= {"xgmi", "pcie", "hypertransport", "qpi", "infiniband", "unknown"}
    cpus = dist.pin_rank_cpus(0, 1)                                        # one rank per host: nothing to slice
    assert cpus is None
