"""The epoch's `opt_times` updates launched back to back with ONE read-back of their logged statistics
(OffRLAlgo.update_per_epoch over `update_deferred` / `resolve_updates`) against the reference's update-by-update loop
(off_rl_algo.py:33-47): same index stream, same launches, so the info dicts and the parameters are bit-identical;
and the one-launch multi-key replay gather against per-key gathers."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _Rec:
    def __init__(self): self.infos = []
    def add_update_info(self, d): self.infos.append(dict(d))
    def add_epoch_info(self, *a, **k): pass
    def log(self, *a): pass
    def finish(self): pass


def _sac_run(deferred, opt_times=5, epochs=2, n_env=256):
    from test_fullsize_offpolicy_gpu import build_cfg3
    pf, qf1, qf2, env, buf, col, agent, _ = build_cfg3(n_env=n_env)
    agent.noise_mode, col.noise_mode = "device", "device"
    agent.logger = _Rec()
    agent.opt_times = opt_times
    torch.manual_seed(5)
    col.train_one_epoch()
    np.random.seed(9)
    for _ in range(epochs):                                              # second epoch: the update graph is replayed
        if deferred:
            agent.update_per_epoch()
        else:
            for _ in range(opt_times):
                agent._one_update()
    eng = agent.engine()
    return agent.logger.infos, eng.flat.cpu().clone(), eng.tflat.cpu().clone(), agent.log_alpha.cpu().clone()


def test_sac_epoch_of_deferred_updates_equals_update_by_update():
    ia, fa, ta, la = _sac_run(False)
    ib, fb, tb, lb = _sac_run(True)
    assert len(ia) == len(ib) == 10
    assert torch.equal(fa, fb) and torch.equal(ta, tb) and torch.equal(la, lb)
    for x, y in zip(ia, ib):
        assert x == y
    assert len({i["Training/qf1_loss"] for i in ib}) == 10               # every slot carries its own update


def test_sac_epoch_as_one_graph_equals_sample_and_enqueue_one_by_one(monkeypatch):
    """TwinSACQ.update_epoch_deferred: the epoch's index sets uploaded as one slab, every update of the one captured graph
    gathering its own set and filing its statistics by the device-resident update count (eager, captured, replayed twice)
    against the per-update {random_batch, enqueue} loop."""
    from torchrl.algo import TwinSACQ
    with monkeypatch.context() as m:                                     # the whole-epoch hook declines: one by one
        m.setattr(TwinSACQ, "update_epoch_deferred", lambda self, count: None)
        ia, fa, ta, la = _sac_run(True, epochs=4)
    ib, fb, tb, lb = _sac_run(True, epochs=4)
    assert len(ia) == len(ib) == 20
    assert torch.equal(fa, fb) and torch.equal(ta, tb) and torch.equal(la, lb)
    for x, y in zip(ia, ib):
        assert x == y
    assert len({i["Training/qf1_loss"] for i in ib}) == 20


class _Later(_Rec):
    """A logger that takes launched-but-not-awaited updates (as torchrl_amd.utils.Logger does)."""
    def __init__(self): super().__init__(); self.later = []
    def add_update_info(self, d): self.drain(); self.infos.append(dict(d))
    def add_update_infos_later(self, resolve): self.later.append(resolve)
    def drain(self):
        later, self.later = self.later, []
        for resolve in later:
            self.infos.extend(dict(d) for d in resolve())


@pytest.mark.parametrize("noise", ["device", "host"])
def test_off_policy_epochs_without_a_host_wait_equal_the_waiting_loop(noise, monkeypatch):
    """RLAlgo.train's order -- collect, update, look at the collector's result -- with the collector's result read on first
    access and the update's info dicts taken later by the logger: three epochs are launched with the only waits being the
    looks at the (already finished) rollout; same episode returns, info dicts and parameters as the loop that reads
    everything back in place."""
    from test_fullsize_offpolicy_gpu import build_cfg3

    def run(lazy):
        pf, qf1, qf2, env, buf, col, agent, _ = build_cfg3(n_env=64)
        col.eager_epoch_result = agent.eager_update_infos = not lazy
        agent.noise_mode = col.noise_mode = noise
        log = agent.logger = _Later()
        agent.opt_times = 3
        torch.manual_seed(5); np.random.seed(9)
        out = []
        for _ in range(4):
            res = col.train_one_epoch()
            assert isinstance(res, dict) != lazy
            agent.update_per_epoch()
            assert bool(log.later) == lazy
            out.append((list(res["train_rewards"]), res["train_epoch_reward"]))
        log.drain()
        eng = agent.engine()
        return out, log.infos, eng.flat.cpu().clone(), eng.tflat.cpu().clone()
    (ra, ia, fa, ta), (rb, ib, fb, tb) = run(False), run(True)
    assert len(ra[0][0]) > 0 and ra == rb and len(ia) == 12 and ia == ib and torch.equal(fa, fb) and torch.equal(ta, tb)


def test_sac_epoch_graph_is_declined_when_its_conditions_do_not_hold(monkeypatch):
    from test_fullsize_offpolicy_gpu import build_cfg3
    pf, qf1, qf2, env, buf, col, agent, _ = build_cfg3(n_env=64)
    agent.noise_mode, col.noise_mode = "host", "host"
    col.train_one_epoch()
    state = np.random.get_state()[1].copy()
    assert agent.update_epoch_deferred(3) is None                        # host noise: one by one
    assert (np.random.get_state()[1] == state).all()                     # ... and no index was drawn
    agent.noise_mode = "device"
    agent.use_soft_update = False
    assert agent.update_epoch_deferred(3) is None                        # hard target updates: one by one
    agent.use_soft_update = True
    handles = agent.update_epoch_deferred(3)
    assert handles is not None and len(handles) == 3 and agent.training_update_num == 3
    infos = agent.resolve_updates(handles)
    assert len({i["Training/qf1_loss"] for i in infos}) == 3


def test_index_slab_gather_picks_the_set_of_the_device_counter():
    from torchrl_amd import _C
    rows, n, sets, nrows = 29, 4, 3, 5
    srcs = [torch.randn(rows, n, 7, device=DEV), torch.randn(rows, n, 1, device=DEV)]
    idx = torch.randint(0, rows, (sets, nrows))
    slab = torch.cat([torch.tensor([10, sets]), idx.reshape(-1)]).to(DEV)
    counter = torch.zeros(4, dtype=torch.float64, device=DEV)
    for step in (10, 11, 12):
        counter[0] = step
        outs = [torch.empty(nrows * n, 7, device=DEV), torch.empty(nrows * n, 1, device=DEV)]
        _C.gather_rows_multi(srcs, slab, outs, slab_counter=counter, n_rows=nrows)
        for s_, o in zip(srcs, outs):
            assert torch.equal(o, s_[idx[step - 10].to(DEV)].reshape(o.shape))
    for step in (9, 13):                                                 # outside the slab: nothing is copied
        counter[0] = step
        outs = [torch.full((nrows * n, 7), 3.0, device=DEV), torch.full((nrows * n, 1), 3.0, device=DEV)]
        _C.gather_rows_multi(srcs, slab, outs, slab_counter=counter, n_rows=nrows)
        assert all(bool((o == 3.0).all()) for o in outs)


def test_moments_launch_files_the_statistics_block_into_its_ring_slot():
    from torchrl_amd import _C
    raw = torch.zeros(160, dtype=torch.uint8, device=DEV)
    sums, mom = raw[:32].view(torch.float64), raw[32:128].view(torch.float64).view(3, 4)
    tail = raw[128:160].view(torch.float32)
    ring = torch.zeros(4, 160, dtype=torch.uint8, device=DEV)
    counter = torch.zeros(4, dtype=torch.float64, device=DEV)
    x = torch.randn(64, 6, device=DEV)
    for done in (1, 2, 6):                                               # slots 0, 1, 5 % 4 = 1
        counter[0] = done
        sums.copy_(torch.arange(4, dtype=torch.float64) + done); tail.fill_(float(done))
        x.mul_(1.5)
        _C.moments_multi([(x, mom[0], 6, 3, 3, -1.0, 1.0), (x, mom[1], 6, 0, 1, -9.0, 9.0), (x, mom[2], 6, 0, 3, -9.0, 9.0)],
                         ring=(raw, ring, counter))
        ref = torch.zeros_like(mom)
        _C.moments_multi([(x, ref[0], 6, 3, 3, -1.0, 1.0), (x, ref[1], 6, 0, 1, -9.0, 9.0), (x, ref[2], 6, 0, 3, -9.0, 9.0)])
        assert torch.equal(mom, ref)
        assert torch.equal(ring[(done - 1) % 4], raw)
    assert bool((ring[2] == 0).all()) and bool((ring[3] == 0).all())
    with pytest.raises(_C.TrlError):                                     # a statistic outside the block cannot be filed
        _C.moments_multi([(x, torch.zeros(4, dtype=torch.float64, device=DEV), 6, 0, 3, -9.0, 9.0)], ring=(raw, ring, counter))


@pytest.mark.parametrize("n_env", [256, 24])                            # B = 1024, and B = 96: a half-empty statistics wave
def test_sac_statistics_riding_on_the_update_s_own_launches_equal_the_separate_launches(n_env, monkeypatch):
    """One rank, soft target updates: the temperature step inside the loss launch, the logged moments from per-wave partials
    of the sampling launch folded by the loss launch, the statistics block filed by the Polyak launch -- against
    trl_sac_alpha_step_f32 / trl_moments_multi_f64 (+ ring) as launches of their own, the route env shards on several ranks
    take (forced here on one rank).  Same arithmetic for everything that feeds back into the update (parameters, targets,
    alpha bit-identical); the moments are summed in another order."""
    from torchrl_amd.algo.off_policy.twin_sac_q import _FusedSAC
    with monkeypatch.context() as m:
        m.setattr(_FusedSAC, "_stats_ride_along", lambda self, soft: False)
        ia, fa, ta, la = _sac_run(True, epochs=3, n_env=n_env)
    ib, fb, tb, lb = _sac_run(True, epochs=3, n_env=n_env)
    assert len(ia) == len(ib) == 15
    assert torch.equal(fa, fb) and torch.equal(ta, tb) and torch.equal(la, lb)
    for x, y in zip(ia, ib):
        assert list(x) == list(y)
        for k in x:
            if k.split("/")[0] in ("log_std", "log_probs", "mean"):
                assert y[k] == pytest.approx(x[k], rel=1e-11, abs=1e-13), k
            else:
                assert x[k] == y[k], k


def test_sac_noise_drawn_inside_the_sampling_launch_equals_the_separate_noise_launches(monkeypatch):
    """trl_sac_samples_f32 with a step state makes update u's two draws from the device-resident update count (2u + 1,
    2u + 2): the values trl_philox_normal_f32 is launched for where the draws are sharded over ranks (forced here on one
    rank) -- parameters, targets, alpha and every logged number agree."""
    from torchrl_amd.algo.off_policy.twin_sac_q import _FusedSAC
    with monkeypatch.context() as m:
        m.setattr(_FusedSAC, "_inline_noise", lambda self: False)
        ia, fa, ta, la = _sac_run(True)
    ib, fb, tb, lb = _sac_run(True)
    assert torch.equal(fa, fb) and torch.equal(ta, tb) and torch.equal(la, lb)
    for x, y in zip(ia, ib):
        assert x == y


@pytest.mark.parametrize("Q", [1, 8])
def test_dqn_epoch_of_deferred_updates_equals_update_by_update(Q):
    from test_fullsize_offpolicy_gpu import build_cfg5
    res = []
    for deferred in (False, True):
        qf, pf, env, buf, col, agent = build_cfg5(Q)
        agent.logger = _Rec()
        agent.opt_times = 4
        np.random.seed(2)
        col.train_one_epoch()
        np.random.seed(3)
        if deferred:
            agent.update_per_epoch()
        else:
            for _ in range(4):
                agent._one_update()
        res.append((agent.logger.infos, agent.engine().flat.cpu().clone(), agent.engine().tflat.cpu().clone()))
    (ia, fa, ta), (ib, fb, tb) = res
    assert len(ia) == len(ib) == 4 and torch.equal(fa, fb) and torch.equal(ta, tb)
    for x, y in zip(ia, ib):
        assert x == y
    assert len({i["Training/qf_loss"] for i in ib}) == 4


@pytest.mark.parametrize("Q,soft", [(1, True), (8, True), (1, False)])
def test_dqn_epoch_as_one_graph_equals_sample_and_enqueue_one_by_one(Q, soft, monkeypatch):
    """DQN.update_epoch_deferred (first epoch one by one -- no update seen yet --, then eager, captured, replayed) against
    the per-update {random_batch, enqueue} loop; with hard target copies the epochs a copy falls into go one by one."""
    from test_fullsize_offpolicy_gpu import build_cfg5
    res = []
    for flag in ("0", "1"):
        qf, pf, env, buf, col, agent = build_cfg5(Q)
        if flag == "0":                                                  # the whole-epoch hook declines: one by one
            agent.update_epoch_deferred = lambda count: None
        agent.logger = _Rec()
        agent.opt_times = 3
        agent.use_soft_update, agent.target_hard_update_period = soft, 7
        np.random.seed(2)
        col.train_one_epoch()
        np.random.seed(3)
        for _ in range(5):
            agent.update_per_epoch()
        eng = agent.engine()
        assert agent.training_update_num == 15 and eng.step_state.cpu()[0].item() == 15
        assert len(eng._graphs) == (1 if flag == "0" else 2)
        res.append((agent.logger.infos, eng.flat.cpu().clone(), eng.tflat.cpu().clone()))
    (ia, fa, ta), (ib, fb, tb) = res
    assert len(ia) == len(ib) == 15 and torch.equal(fa, fb) and torch.equal(ta, tb)
    for x, y in zip(ia, ib):
        assert x == y
    assert len({i["Training/qf_loss"] for i in ib}) == 15


@pytest.mark.parametrize("Q", [1, 5])
def test_loss_launch_with_stored_float_actions_files_its_sums(Q):
    from torchrl_amd import _C
    torch.manual_seed(Q)
    B, A = 96, 6
    q, qn = torch.randn(B, A * Q, device=DEV), torch.randn(B, A * Q, device=DEV)
    acts = torch.randint(0, A, (B,), device=DEV)
    rew, term = torch.randn(B, device=DEV), (torch.rand(B, device=DEV) < 0.1).float()
    s_i, s_f = torch.zeros(3, dtype=torch.float64, device=DEV), torch.zeros(3, dtype=torch.float64, device=DEV)
    ring = torch.zeros(4, 3, dtype=torch.float64, device=DEV)
    counter = torch.tensor([6.0, 1.0, 1.0, 0.0], dtype=torch.float64, device=DEV)    # 6 updates done: row 6 % 4
    if Q == 1:
        d_i = _C.dqn_td_loss(q, acts, qn, rew, term, 0.99, s_i)
        d_f = _C.dqn_td_loss(q, acts.float(), qn, rew, term, 0.99, s_f, ring=(ring, counter))
    else:
        d_i = _C.quantile_huber(q, acts, qn, rew, term, 0.99, A, Q, s_i)
        d_f = _C.quantile_huber(q, acts.float(), qn, rew, term, 0.99, A, Q, s_f, ring=(ring, counter))
    assert torch.equal(d_i, d_f) and torch.equal(s_i, s_f) and torch.equal(ring[2], s_f)
    assert bool((ring[[0, 1, 3]] == 0).all())
    # a stored action outside [0, A) (or NaN) is clamped inside the kernel instead of indexing out of bounds
    bad = acts.float().clone()
    bad[0], bad[1], bad[2] = float("nan"), -3.0, 1e9
    fixed = acts.clone()
    fixed[0], fixed[1], fixed[2] = 0, 0, A - 1
    if Q == 1:
        assert torch.equal(_C.dqn_td_loss(q, bad, qn, rew, term, 0.99, s_f), _C.dqn_td_loss(q, fixed, qn, rew, term, 0.99, s_i))
    else:
        assert torch.equal(_C.quantile_huber(q, bad, qn, rew, term, 0.99, A, Q, s_f),
                           _C.quantile_huber(q, fixed, qn, rew, term, 0.99, A, Q, s_i))
    assert torch.equal(s_i, s_f)


@pytest.mark.parametrize("Q", [1, 8])
def test_dqn_graph_replayed_updates_equal_eager_launches(Q, monkeypatch):
    """Updates 3+ of a configuration replay a captured HIP graph (fixed-address inputs filled by the replay gather, Adam
    step count on the device): same launches, so parameters and logged statistics equal the eager sequence bit for bit."""
    from test_fullsize_offpolicy_gpu import build_cfg5
    res = []
    for no_graph in ("1", "0"):
        monkeypatch.setenv("TRL_NO_GRAPH", no_graph)
        qf, pf, env, buf, col, agent = build_cfg5(Q)
        agent.logger = _Rec()
        agent.opt_times = 3
        np.random.seed(2)
        col.train_one_epoch()
        np.random.seed(3)
        for _ in range(2):
            agent.update_per_epoch()
        eng = agent.engine()
        assert len(eng._graphs) == (0 if no_graph == "1" else 1) and eng.step_state.cpu()[0].item() == 6
        assert eng.static_batch()["obs"].dtype == torch.uint8
        res.append((agent.logger.infos, eng.flat.cpu().clone(), eng.tflat.cpu().clone()))
    (ia, fa, ta), (ib, fb, tb) = res
    assert len(ia) == len(ib) == 6 and torch.equal(fa, fb) and torch.equal(ta, tb)
    for x, y in zip(ia, ib):
        assert x == y


@pytest.mark.parametrize("kind", ["ddpg", "td3"])
def test_ddpg_td3_epoch_of_deferred_updates_equals_update_by_update(kind):
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.algo import DDPG, TD3
    from torchrl.collector import VecCollector
    from torchrl.env.synth import SynthVecEnv
    from torchrl.replay_buffers import BaseReplayBuffer
    N, T, dev = 32, 12, torch.device(DEV)
    res = []
    for deferred in (False, True):
        torch.manual_seed(4)
        net = dict(hidden_shapes=[64, 64], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.ReLU)
        pf = policies.FixGuassianContPolicy(input_shape=17, output_shape=6, tanh_action=True, norm_std_explore=0.1, **net)
        env, ev = SynthVecEnv(N, horizon=5, device=dev), SynthVecEnv(N, horizon=5, device=dev)
        env.seed(2)
        buf = BaseReplayBuffer(N * T, env_nums=N)
        col = VecCollector(env=env, eval_env=ev, pf=pf, replay_buffer=buf, device=dev, epoch_frames=N * T,
                           max_episode_frames=1000, eval_episodes=1)
        col.train_one_epoch()
        log = _Rec()
        kw = dict(env=env, replay_buffer=buf, collector=col, logger=log, grad_clip=1.0, discount=0.99, num_epochs=1,
                  batch_size=N * 4, device=dev, save_dir=None, tau=0.005, use_soft_update=True, opt_times=5)
        q = lambda: networks.QNet(input_shape=23, output_shape=1, **net)
        agent = DDPG(pf=pf, qf=q(), plr=3e-4, qlr=1e-3, **kw) if kind == "ddpg" else \
            TD3(pf=pf, qf1=q(), qf2=q(), plr=3e-4, qlr=1e-3, noise_mode="device", **kw)
        np.random.seed(6)
        for _ in range(2):                                                   # the second epoch replays the captured graphs
            if deferred:
                agent.update_per_epoch()
            else:
                for _ in range(5):
                    agent._one_update()
        res.append((log.infos, agent.engine().flat.cpu().clone(), agent.engine().tflat.cpu().clone()))
    (ia, fa, ta), (ib, fb, tb) = res
    assert len(ia) == len(ib) == 10 and torch.equal(fa, fb) and torch.equal(ta, tb)
    for x, y in zip(ia, ib):
        assert x == y
    if kind == "td3":                                                        # delayed policy steps keep their own keys
        assert {len(i) for i in ib} == {len(ib[0]), len(ib[1])} and len(ib[0]) != len(ib[1])


def test_more_pending_updates_than_ring_slots_still_resolve_in_order():
    from test_fullsize_offpolicy_gpu import build_cfg3
    pf, qf1, qf2, env, buf, col, agent, _ = build_cfg3(n_env=64)
    agent.noise_mode, col.noise_mode = "device", "device"
    torch.manual_seed(5)
    col.train_one_epoch()
    np.random.seed(1)
    handles = [agent.update_deferred(agent._sample()) for _ in range(70)]          # ring = 64 slots
    infos = agent.resolve_updates(handles)
    assert len(infos) == 70 and len({id(h[0]) for h in handles}) == 2
    assert len({i["Training/qf1_loss"] for i in infos}) == 70
    one = agent.update(agent._sample())                                  # the plain call keeps working afterwards
    assert np.isfinite(list(one.values())).all()


def test_multi_key_gather_equals_per_key_gathers():
    from torchrl_amd import _C
    from torchrl_amd.replay_buffers import BaseReplayBuffer
    rows, n = 37, 8
    srcs = [torch.randn(rows, n, 17, device=DEV), torch.randn(rows, n, 17, device=DEV), torch.randn(rows, n, 6, device=DEV),
            torch.randn(rows, n, 1, device=DEV), (torch.rand(rows, n, 3, 5, device=DEV) * 255).to(torch.uint8)]
    idx = torch.tensor([36, 0, 5, 5, 12], device=DEV)
    outs = [torch.empty((5 * n,) + tuple(s.shape[2:]), dtype=s.dtype, device=DEV) for s in srcs]
    _C.gather_rows_multi(srcs, idx, outs)
    for s, o in zip(srcs, outs):
        assert torch.equal(o, s[idx].reshape(o.shape))
    # out-of-range rows are skipped, never copied from wild memory
    outs2 = [torch.full_like(o, 7) for o in outs]
    _C.gather_rows_multi(srcs, torch.tensor([99, 1, -1, 2, 3], device=DEV), outs2)
    assert torch.equal(outs2[0][:n], torch.full_like(outs2[0][:n], 7)) and torch.equal(outs2[0][n:2 * n], srcs[0][1])
    with pytest.raises(_C.TrlError):
        _C.gather_rows_multi(srcs[:2], idx, [outs[0], outs[2]])
    # the replay buffer's sample uses it: same batch as key-by-key gathers, into caller-owned tensors too
    buf = BaseReplayBuffer(rows * n, env_nums=n, device=DEV)
    for key, s in zip(("obs", "next_obs", "acts", "rewards"), srcs):
        setattr(buf, "_" + key, s); buf._keys.append(key)
    buf._size = rows
    np.random.seed(4)
    got = buf.random_batch(3 * n, ["obs", "next_obs", "acts", "rewards"])
    np.random.seed(4)
    pick = torch.as_tensor(np.random.randint(0, rows, 3)).to(DEV)
    for key, s in zip(("obs", "next_obs", "acts", "rewards"), srcs):
        assert torch.equal(got[key], s[pick].reshape(got[key].shape)) and got[key].shape[0] == 3 * n
    static = {"obs": torch.empty(3 * n, 17, device=DEV), "acts": torch.empty(3 * n, 6, device=DEV)}
    np.random.seed(4)
    again = buf.random_batch(3 * n, ["obs", "next_obs", "acts", "rewards"], out=static)
    assert again["obs"] is static["obs"] and torch.equal(static["acts"], got["acts"]) and torch.equal(again["rewards"], got["rewards"])
