"""The reference's exploration-noise stream drawn one rollout ahead (VecOnPolicyCollector(prefetch_noise=True),
torchrl_amd/collector/on_policy.py::_NoisePrefetcher): the CPU generator's values of
torchrl/policies/distribution.py:60-76 in the same order, only earlier -- so everything downstream must be bit-identical
to the un-prefetched host mode."""
import numpy as np
import pytest
import torch

from test_product_gpu import build

pytestmark = pytest.mark.gpu
N, T, HORIZON, MAX_FRAMES, B = 64, 16, 12, 9, 256
KEYS = ("obs", "next_obs", "acts", "values", "rewards", "terminals", "time_limits", "old_logp")


def _run(prefetch, iterations=3, reseed_at=None, stop_after=None, transport=None):
    torch.manual_seed(11)
    pf, vf, env, buf, col, agent, logger = build(None, "", N, T, HORIZON, MAX_FRAMES, B, 5)
    col.prefetch_noise = prefetch
    if transport is not None:                                          # pin the block's way to the device
        from torchrl_amd.collector.on_policy import _NoisePrefetcher
        pre = col._prefetcher = _NoisePrefetcher(env.device)
        pre.staged, pre.carry = transport != "stream", transport == "carried"
        pre.wait_for_draw = transport == "carried"
    torch.manual_seed(5)
    np.random.seed(5)
    snaps = []
    for it in range(iterations):
        if reseed_at == it:
            torch.manual_seed(77)                                      # the generator is touched between two rollouts
        col.train_one_epoch()
        snaps.append({k: getattr(buf, "_" + k).clone() for k in KEYS})
        agent.current_epoch = it
        agent.update_per_epoch()
        if stop_after == it:
            col.stop_noise_prefetch()
    torch.cuda.synchronize()
    tail = torch.randn(7)                                              # where the CPU stream stands afterwards
    col.stop_noise_prefetch()
    return snaps, pf.flat_params().clone(), vf.flat_params().clone(), tail, col


def test_prefetched_noise_equals_the_in_place_draws():
    ref, pf0, vf0, _, _ = _run(False)
    got, pf1, vf1, _, col = _run(True)
    assert col._prefetcher is not None                                 # the prefetcher did carry the rollouts
    for a, b in zip(ref, got):
        for k in KEYS:
            assert torch.equal(a[k], b[k]), k
    assert torch.equal(pf0, pf1) and torch.equal(vf0, vf1)


def test_generator_seeded_between_rollouts_drops_the_prefetched_block():
    ref, pf0, _, tail0, _ = _run(False, reseed_at=2)
    got, pf1, _, _, _ = _run(True, reseed_at=2)
    for a, b in zip(ref, got):
        for k in KEYS:
            assert torch.equal(a[k], b[k]), k
    assert torch.equal(pf0, pf1)


def test_stopping_the_prefetch_rewinds_the_generator():
    """An abandoned speculative block leaves no trace: after stop_noise_prefetch() the CPU stream stands where the
    un-prefetched run left it."""
    _, _, _, tail0, _ = _run(False, iterations=2)
    _, _, _, tail1, _ = _run(True, iterations=2, stop_after=1)
    assert torch.equal(tail0, tail1)


@pytest.mark.parametrize("transport", ["carried", "staged", "stream"])
def test_every_way_of_the_block_to_the_device_gives_the_same_rollouts(transport):
    """The prefetched block reaches the device (a) staged by extra workgroups of the previous rollout launch, (b) by a
    staging kernel on a side stream, (c) by a copy command on the rollout's stream -- (a) falls back to (b) whenever the
    host is not ahead of the device.  Each way pinned: same buffers and parameters as the in-place draws, also across a
    re-seed of the generator (which drops a block that is already on its way)."""
    ref, pf0, vf0, _, _ = _run(False, iterations=5, reseed_at=3)
    got, pf1, vf1, _, col = _run(True, iterations=5, reseed_at=3, transport=transport)
    for a, b in zip(ref, got):
        for k in KEYS:
            assert torch.equal(a[k], b[k]), (transport, k)
    assert torch.equal(pf0, pf1) and torch.equal(vf0, vf1)
    counts = col._prefetcher.transport_counts
    assert counts[transport] >= 2, counts                                # the pinned way was really taken
    if transport != "carried":
        assert counts["carried"] == 0
