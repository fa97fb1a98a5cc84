"""Multi-GPU glue: one process per GPU, envs sharded by index, RCCL collectives
through torch.distributed (backend "nccl" == RCCL on ROCm; "gloo" in CPU tests).

The reference has no distributed path.  What makes sharding exact here is its
minibatch definition -- `B // N` time rows x ALL N envs
(torchrl/replay_buffers/on_policy.py:75-88): when every rank draws the same row
indices (same numpy seed) and owns envs [rank*N/G, (rank+1)*N/G), the union of
the rank-local minibatches IS the single-process minibatch.  So the data path
needs no collective; only three small reductions exist:
  C1  flat gradient SUM (per-sample grads already carry 1/n_global),
  C2  advantage statistics {sum, sumsq} SUM and {max, -min} MAX, once per epoch,
  C3  logging statistics, once per epoch.
"""
import os

import torch
import torch.distributed as td

# TRL_FORCE_COLLECTIVES=1 issues the collectives even at world size 1 (single-GPU smoke test of the
# RCCL call path: dtypes, in-place views, stream ordering)
_FORCE = os.environ.get("TRL_FORCE_COLLECTIVES", "0") == "1"


def initialized():
    return td.is_available() and td.is_initialized()


def world_size():
    return td.get_world_size() if initialized() else 1


def rank():
    return td.get_rank() if initialized() else 0


def shard(total_envs, world=None, r=None):
    """(offset, count) of the env-index block owned by rank r."""
    world = world_size() if world is None else world
    r = rank() if r is None else r
    if total_envs % world != 0:
        raise ValueError("total env count %d is not divisible by world size %d" % (total_envs, world))
    per = total_envs // world
    return r * per, per


def _active():
    return initialized() and (world_size() > 1 or _FORCE)


def collectives_active():
    """True when gradients / statistics are exchanged between ranks (world size > 1, or forced)."""
    return _active()


def all_reduce_sum_(t):
    if _active():
        td.all_reduce(t, op=td.ReduceOp.SUM)
    return t


def all_reduce_max_(t):
    if _active():
        td.all_reduce(t, op=td.ReduceOp.MAX)
    return t


def reduce_adv_raw_(raw):
    """raw: (K, 4) float64 {sum, sumsq, max, -min} -> global statistics (C2)."""
    if _active():
        s = raw[:, :2].contiguous()
        m = raw[:, 2:].contiguous()
        all_reduce_sum_(s)
        all_reduce_max_(m)
        raw[:, :2] = s
        raw[:, 2:] = m
    return raw


# columns of the per-update info rows (include/trl_hip.h, trl_ppo_reduce_f32)
# sums: surrogate, log-prob sum / sum of squares, value loss, value-prediction sum / sum of squares;
# maxima: max / -min of log-prob, ratio and value prediction.  Slots 8..11 and 16..19 derive from logstd (replicated).
INFO_SUM_COLS = [0, 1, 2, 7, 12, 13]
INFO_MAX_COLS = [3, 4, 5, 6, 14, 15]


def reduce_info_(info):
    """info: (K, 24) float64 per-update statistics -> global (C3)."""
    if _active():
        s = info[:, INFO_SUM_COLS].contiguous()
        m = info[:, INFO_MAX_COLS].contiguous()
        all_reduce_sum_(s)
        all_reduce_max_(m)
        info[:, INFO_SUM_COLS] = s
        info[:, INFO_MAX_COLS] = m
    return info


def all_gather_cat(t):
    """Rows of every rank's tensor, concatenated in rank order (identity at world size 1)."""
    if not _active():
        return t
    parts = [torch.empty_like(t) for _ in range(world_size())]
    td.all_gather(parts, t.contiguous())
    return torch.cat(parts, dim=0)


def gather_env_shards(t, n_local):
    """(rows * n_local, feat) minibatch shard of every rank -> the (rows * n_total, feat) minibatch a single process
    would hold: row-major over (sampled time row, env), envs in rank order.  Identity at world size 1."""
    if not _active():
        return t
    feat = t.shape[1:]
    rows = t.shape[0] // n_local
    parts = [torch.empty_like(t) for _ in range(world_size())]
    td.all_gather(parts, t.contiguous())
    whole = torch.cat([p.view(rows, n_local, *feat) for p in parts], dim=1)
    return whole.reshape(rows * whole.shape[1], *feat).contiguous()


def shard_rows_of_global(make_global, rows, n_local, feat, device):
    """The (rows * n_local, feat) block of this rank out of a tensor generated for ALL ranks: `make_global(rows * n_total,
    feat)` is laid out (rows, n_total, feat) -- a sampled time row x all envs -- and the rank owns envs
    [rank * n_local, (rank + 1) * n_local).  Every rank generates the same global tensor (same seeds), so the union
    over ranks is exactly what a single process would have drawn."""
    w, r = world_size(), rank()
    if w == 1:
        return make_global(rows * n_local, feat)
    g = make_global(rows * n_local * w, feat).view(rows, n_local * w, feat)
    return g[:, r * n_local:(r + 1) * n_local, :].reshape(rows * n_local, feat).to(device).contiguous()
