"""Multi-GPU path on CPU: two gloo processes exercise env sharding, the collective helpers and
the sharded-minibatch identity (union of rank-local minibatches == single-process minibatch;
SUM of rank gradients carrying 1/n_global == full-batch gradient), using the CPU oracle for the math."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    import torch.distributed as td
    from torchrl_amd import dist
    from oracle import nets
    td.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        assert dist.initialized() and dist.world_size() == world and dist.rank() == rank
        N_total, T, rows_mb, D, A = 8, 6, 3, 17, 6
        off, cnt = dist.shard(N_total)
        assert (off, cnt) == (rank * 4, 4)
        with pytest.raises(ValueError):
            dist.shard(7)
        # identical index streams on every rank (same numpy seed) -- the sharding contract
        np.random.seed(0)
        idx = np.random.permutation(T)[:rows_mb]
        gathered = [None] * world
        td.all_gather_object(gathered, idx.tolist())
        assert gathered[0] == gathered[1]

        rs = np.random.RandomState(5)
        full = {"obs": rs.randn(T, N_total, D).astype(np.float32),
                "acts": (np.tanh(rs.randn(T, N_total, A)) * 0.9).astype(np.float32),
                "advs": rs.randn(T, N_total, 1).astype(np.float32)}
        gen = torch.Generator().manual_seed(3)
        pf = nets.init_mlp(D, [64, 64], A, generator=gen)
        pf = [p.requires_grad_(True) for p in pf]
        ls = torch.full((A,), -1.0, requires_grad=True)

        def loss_sum(obs, acts, advn):                      # un-normalised sum; 1/n_global applied below
            out = nets.policy_update_terms(torch.tensor(obs), torch.tensor(acts), pf, ls)
            return -(torch.exp(out["log_prob"] - out["log_prob"].detach()) * torch.tensor(advn)).sum()

        # C2: advantage statistics {sum, sumsq, max, -min} of the LOCAL shard, reduced
        loc = {k: v[idx][:, off:off + cnt] for k, v in full.items()}
        a = loc["advs"].astype(np.float64).reshape(-1)
        raw = torch.tensor([[a.sum(), (a * a).sum(), a.max(), -a.min()]], dtype=torch.float64)
        dist.reduce_adv_raw_(raw)
        fa = full["advs"][idx].astype(np.float64).reshape(-1)
        n = fa.size
        np.testing.assert_allclose(raw[0].numpy(), [fa.sum(), (fa * fa).sum(), fa.max(), -fa.min()], rtol=1e-12)
        mean, std = raw[0, 0].item() / n, np.sqrt((raw[0, 1].item() - raw[0, 0].item() ** 2 / n) / (n - 1))
        assert abs(std - torch.tensor(fa).std().item()) < 1e-9

        # C1: SUM of rank gradients (each carrying 1/n_global) == gradient of the full-batch mean loss
        advn_loc = ((loc["advs"] - mean) / (std + 1e-5)).astype(np.float32).reshape(-1, 1)
        g_loc = torch.autograd.grad(loss_sum(loc["obs"].reshape(-1, D), loc["acts"].reshape(-1, A), advn_loc) / n,
                                    pf + [ls])
        flat = torch.cat([g.reshape(-1) for g in g_loc])
        dist.all_reduce_sum_(flat)
        advn_full = ((full["advs"][idx] - mean) / (std + 1e-5)).astype(np.float32).reshape(-1, 1)
        g_full = torch.autograd.grad(loss_sum(full["obs"][idx].reshape(-1, D), full["acts"][idx].reshape(-1, A),
                                              advn_full) / n, pf + [ls])
        want = torch.cat([g.reshape(-1) for g in g_full])
        assert (flat - want).abs().max().item() < 1e-6 * max(1.0, want.abs().max().item())

        # C3: logging statistics
        info = torch.zeros(2, 24, dtype=torch.float64)
        info[:, dist.INFO_SUM_COLS] = float(rank + 1)
        info[:, dist.INFO_MAX_COLS] = float(rank)
        info[:, 8] = 7.0                                   # log_std columns are rank-identical: untouched
        dist.reduce_info_(info)
        assert torch.all(info[:, dist.INFO_SUM_COLS] == 3.0) and torch.all(info[:, dist.INFO_MAX_COLS] == 1.0)
        assert torch.all(info[:, 8] == 7.0)
        t = torch.tensor([float(rank)])
        dist.all_reduce_max_(t)
        assert t.item() == world - 1

        # observation normaliser: SUM of the shards' batch moments {sum x, sum x^2, n}, then the
        # same Chan merge on every rank == the single-process update on the full batch (what
        # torchrl_amd.env.base_wrapper.Normalizer does between trl_norm_batch_moments_f64 and trl_norm_merge_f64)
        from oracle.normalizer import NormalizerOracle, update_mean_var_count
        rs2 = np.random.RandomState(7)
        state = (np.zeros(D), np.ones(D), 1e-4)
        ref = NormalizerOracle((D,))
        for step in range(3):
            xfull = rs2.randn(N_total, D) * (1 + step) + 0.2 * step
            xloc = xfull[off:off + cnt]
            sums = torch.tensor(np.concatenate([xloc.sum(0), (xloc * xloc).sum(0), [float(cnt)]]))
            dist.all_reduce_sum_(sums)
            sums = sums.numpy()
            n_glob = sums[-1]
            bmean = sums[:D] / n_glob
            bvar = np.maximum(sums[D:2 * D] / n_glob - bmean * bmean, 0.0)
            state = update_mean_var_count(*state, bmean, bvar, n_glob)
            ref.update_estimate(xfull)
            np.testing.assert_allclose(state[0], ref._mean, rtol=1e-12, atol=1e-13)
            np.testing.assert_allclose(state[1], ref._var, rtol=1e-11, atol=1e-13)
            assert abs(state[2] - ref._count) < 1e-9
        # noise / host-draw sharding: every rank generates the draw for ALL envs and keeps its own env block, so the
        # union over ranks is what one process would have drawn (off-policy engines, per-step collection, eps-greedy)
        rows, n_loc, feat = 3, 4, 2
        mk = lambda m, f: torch.manual_seed(11) and torch.randn(m, f)
        mine = dist.shard_rows_of_global(mk, rows, n_loc, feat, "cpu")
        whole = mk(rows * n_loc * world, feat).view(rows, n_loc * world, feat)
        assert torch.equal(mine, whole[:, rank * n_loc:(rank + 1) * n_loc].reshape(rows * n_loc, feat))
        # ... and gather_env_shards is its inverse: the shards reassemble into the single-process minibatch (V-MPO, TRPO)
        assert torch.equal(dist.gather_env_shards(mine, n_loc), whole.reshape(rows * n_loc * world, feat))
        assert torch.equal(dist.gather_env_shards(mine[:, 0].contiguous().view(-1, 1), n_loc).view(-1),
                           whole[:, :, 0].reshape(-1))
        cat = dist.all_gather_cat(mine)
        assert cat.shape == (world * rows * n_loc, feat) and torch.equal(cat[rank * rows * n_loc:(rank + 1) * rows * n_loc], mine)
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        td.destroy_process_group()


def test_two_rank_gloo_sharding_and_collectives(tmp_path):
    world = 2
    try:
        mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    except Exception as exc:                                         # noqa: BLE001 -- the port was free when picked, not reserved
        if "EADDRINUSE" not in str(exc) and "address already in use" not in str(exc):
            raise
        mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % r)) for r in range(world))


def test_single_process_sharding_helpers_are_identities():
    from torchrl_amd import dist
    x = torch.arange(12.0).view(6, 2)
    assert dist.all_gather_cat(x) is x
    assert torch.equal(dist.shard_rows_of_global(lambda m, f: torch.arange(float(m * f)).view(m, f), 3, 2, 2, "cpu"), x)


def test_single_process_helpers_are_noops():
    from torchrl_amd import dist
    assert not dist.initialized() and dist.world_size() == 1 and dist.rank() == 0
    assert dist.shard(2048) == (0, 2048)
    t = torch.ones(3)
    assert dist.all_reduce_sum_(t) is t and torch.all(t == 1)
    raw = torch.ones(2, 4, dtype=torch.float64)
    assert torch.all(dist.reduce_adv_raw_(raw) == 1)
