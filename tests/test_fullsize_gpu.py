"""BASELINE-size (cfg 2: 2048 envs x 128 steps, B = 65 536) checks through size-independent
properties, and the multi-GPU sharding identity emulated on one GPU."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
N, T, B = 2048, 128, 65536


def make(n_env, offset=0, total=None, seed=0, noise="device"):
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.algo import PPO
    from torchrl.collector.on_policy import VecOnPolicyCollector
    from torchrl.env.synth import SynthVecEnv
    from torchrl.replay_buffers.on_policy import OnPolicyReplayBuffer
    torch.manual_seed(11)
    net = dict(hidden_shapes=[64, 64], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.Tanh)
    pf = policies.GuassianContPolicyBasicBias(input_shape=17, output_shape=6, tanh_action=True, **net)
    vf = networks.Net(input_shape=(17,), output_shape=1, **net)
    env = SynthVecEnv(n_env, horizon=50, device=DEV, index_offset=offset, total_env_nums=total)
    ev = SynthVecEnv(n_env, horizon=50, device=DEV, index_offset=offset, total_env_nums=total)
    env.seed(seed)
    buf = OnPolicyReplayBuffer(n_env * T, env_nums=n_env, time_limit_filter=True)
    col = VecOnPolicyCollector(vf, env=env, eval_env=ev, pf=pf, replay_buffer=buf, device=DEV, epoch_frames=n_env * T,
                               max_episode_frames=40, noise_mode=noise)

    class Log:
        infos = []
        def add_update_info(self, d): self.infos.append(d)
        def add_epoch_info(self, *a, **k): pass
        def log(self, *a): pass
        def finish(self): pass
    agent = PPO(pf=pf, vf=vf, plr=3e-4, vlr=3e-4, clip_para=0.2, opt_epochs=1, tau=0.95, shuffle=True,
                entropy_coeff=0.005, discount=0.99, num_epochs=10, batch_size=B * n_env // N, gae=True, env=env,
                replay_buffer=buf, collector=col, logger=Log(), device=DEV, save_dir=None)
    return pf, vf, env, buf, col, agent


def test_full_size_rollout_is_env_shard_invariant():
    """Envs are independent and keyed by their GLOBAL index: a 1024-env shard at offset 1024 of a
    2048-env logical vector env reproduces columns [1024, 2048) of the single-process rollout
    (the multi-GPU partition of SURVEY.md section 8(e), emulated on one GPU)."""
    _, _, _, full, col, _ = make(N)
    col.train_one_epoch()
    for off in (0, 1024):
        _, _, _, part, colp, _ = make(1024, offset=off, total=N)
        colp.train_one_epoch()
        for k in ("obs", "next_obs", "acts", "values", "rewards", "terminals", "time_limits", "old_logp"):
            a, b = getattr(full, "_" + k)[:, off:off + 1024], getattr(part, "_" + k)
            assert torch.equal(a, b), (k, off, (a - b).abs().max().item())
    assert float(full._terminals.sum()) > 0


def test_full_size_gae_properties():
    """Linearity in the rewards and the telescoping identity ret - adv == values, at 128 x 2048."""
    from torchrl_amd import _C
    g = torch.Generator(device="cpu").manual_seed(0)
    r1, r2, v = (torch.randn(T, N, generator=g).to(DEV) for _ in range(3))
    d = (torch.rand(T, N, generator=g) < 0.02).float().to(DEV)
    tl = ((torch.rand(T, N, generator=g) < 0.5).float().to(DEV)) * d
    zeros, lv = torch.zeros(T, N, device=DEV), torch.randn(N, generator=g).to(DEV)

    def run(r, vals, lastv):
        adv, ret = torch.empty(T, N, device=DEV), torch.empty(T, N, device=DEV)
        _C.gae(r, vals, d, tl, lastv, adv, ret, 0.99, 0.95, 1)
        return adv, ret
    a1, ret1 = run(r1, v, lv)
    a2, _ = run(r2, zeros, torch.zeros(N, device=DEV))
    a12, _ = run(r1 + r2, v, lv)
    assert (a12 - (a1 + a2)).abs().max().item() < 2e-4          # A(r1 + r2, V) = A(r1, V) + A(r2, 0)
    assert (ret1 - a1 - v).abs().max().item() < 1e-5            # estimate_returns = advs + values
    flagged = tl > 0
    assert float(a1[flagged].abs().max()) == 0.0                # time-limit filter zeroes the advantage there


def test_full_size_minibatch_gradient_partition_invariance():
    """One cfg-2 minibatch (32 rows x 2048 envs): the folded gradient must not depend on how the
    tiles are spread over workgroups, and the two env shards' gradients (each carrying 1/n_global)
    sum to the full gradient -- the all-reduce identity of the multi-GPU path."""
    from torchrl_amd import _C
    pf, vf, env, buf, col, agent = make(N)
    col.train_one_epoch()
    agent.current_epoch = 0
    agent.process_epoch_samples()
    agent._fill_old_logp()
    eng = agent.engine()
    np.random.seed(3)
    rows = buf.epoch_row_indices(B, True)[0]                      # (32,) time rows
    t = {"obs": buf._obs, "acts": buf._acts, "advs": buf._advs, "rets": buf._estimate_returns,
         "old_values": buf._values, "old_logp": buf._old_logp}
    idx = torch.from_numpy(rows).to(DEV)
    raw = torch.zeros(1, 4, dtype=torch.float64, device=DEV)
    _C.adv_stats(buf._advs.reshape(T, N), idx.reshape(1, -1), raw)

    def grad(tensors, n_env, n_wg, n_global):
        g = _C.PpoBatchArgs()
        for k in ("obs", "acts", "advs", "rets", "old_values", "old_logp"):
            setattr(g, k, tensors[k].data_ptr())
        g.row_idx, g.rows_mb, g.N = idx.data_ptr(), len(rows), n_env
        g.adv_raw, g.n_global = raw.data_ptr(), float(n_global)
        g.pf_params, g.vf_params = eng.flat.data_ptr(), eng.flat.data_ptr() + 4 * eng.P_pf
        g.D, g.H, g.A, g.act = 17, 64, 6, _C.ACT_TANH
        g.clip_para, g.entropy_coeff, g.clipped_value_loss, g.tanh_action = 0.2, 0.005, 0, 1
        partial = torch.zeros(n_wg, eng.p_stride, device=DEV)
        scal = torch.zeros(n_wg, 8, dtype=torch.float64, device=DEV)
        g.partial, g.scal_partial, g.n_wg = partial.data_ptr(), scal.data_ptr(), n_wg
        _C.ppo_minibatch_grad(g, DEV)
        out, info = torch.zeros_like(eng.grads), torch.zeros(24, dtype=torch.float64, device=DEV)
        _C.ppo_reduce(partial, scal, n_wg, 17, 64, 6, out, info, eng.flat)
        return out, info
    g256, i256 = grad(t, N, 256, B)
    g64, i64 = grad(t, N, 64, B)
    scale = g256.abs().max().item()
    assert (g256 - g64).abs().max().item() < 2e-6 * max(1.0, scale)
    # surrogate SUM over 65 536 O(1) terms that cancel (normalised advantages): compare per sample
    assert abs(i256[0].item() - i64[0].item()) / B < 1e-9
    g_again, _ = grad(t, N, 256, B)
    assert torch.equal(g256, g_again)                              # deterministic fold
    halves = []
    for off in (0, 1024):
        th = {k: v[:, off:off + 1024].contiguous() for k, v in t.items()}
        halves.append(grad(th, 1024, 128, B)[0])
    assert ((halves[0] + halves[1]) - g256).abs().max().item() < 2e-6 * max(1.0, scale)


def test_full_size_epoch_runs_and_statistics_are_consistent():
    pf, vf, env, buf, col, agent = make(N)
    col.train_one_epoch()
    agent.current_epoch = 0
    np.random.seed(0)
    agent.update_per_epoch()
    infos = agent.logger.infos[-4:]
    assert len(infos) == 4 and all(np.isfinite(list(i.values())).all() for i in infos)
    # first minibatch: same parameters as at collection time -> ratio is 1 up to the different fp32
    # summation order of the collector (32x32x2 MFMA) and the gradient kernel (16x16x4 MFMA)
    assert abs(infos[0]["ratio/max"] - 1.0) < 2e-5 and abs(infos[0]["ratio/min"] - 1.0) < 2e-5
    for i in infos:
        assert i["advs/min"] <= i["advs/mean"] <= i["advs/max"] and i["logprob/min"] <= i["logprob/mean"] <= i["logprob/max"]


def test_full_size_headline_configuration_prefetched_parallel_reference_noise():
    """The configuration bench.py's headline times (VERDICT r03 missing 6 / weak 1): N = 2048, T = 128, the reference's
    exploration-noise stream prefetched one rollout ahead, the 1.57 M-value block drawn by 8 host threads from derived
    engine states (above noise.MIN_PARALLEL) and carried to the device by the previous rollout launch -- against the
    reference's literal op sequence, one `torch.randn(N, A)` per vector step on the calling thread
    (torchrl/policies/distribution.py:60-76).  All 8 rollout buffers of every iteration and the parameters after 3
    iterations (12 updates of B = 65 536) must be torch.equal."""
    from torchrl_amd.collector import noise
    from torchrl_amd.collector.on_policy import _NoisePrefetcher
    keys = ("obs", "next_obs", "acts", "values", "rewards", "terminals", "time_limits", "old_logp")
    assert N * T * 6 >= noise.MIN_PARALLEL

    def run(prefetch):
        pf, vf, env, buf, col, agent = make(N, noise="host")
        col.prefetch_noise = prefetch
        old = noise._checked
        if prefetch:
            pre = col._prefetcher = _NoisePrefetcher(env.device)
            pre.draw_threads = 8
            pre.wait_for_draw = True                                   # the host is ahead of the device: blocks ride on launches
        else:
            noise._checked = False                                     # per-step torch.randn(N, A), the reference's draws
        try:
            torch.manual_seed(5)
            np.random.seed(5)
            snaps = []
            for it in range(3):
                col.train_one_epoch()
                snaps.append({k: getattr(buf, "_" + k).clone() for k in keys})
                agent.current_epoch = it
                agent.update_per_epoch()
            torch.cuda.synchronize()
            col.stop_noise_prefetch()
            tail = torch.randn(9)
        finally:
            noise._checked = old
        return snaps, pf.flat_params().clone(), vf.flat_params().clone(), tail, col

    par0 = noise.STATS["parallel_blocks"]
    ref, pf0, vf0, tail0, _ = run(False)
    assert noise.STATS["parallel_blocks"] == par0                      # the reference run drew step by step
    got, pf1, vf1, tail1, col = run(True)
    for it, (a, b) in enumerate(zip(ref, got)):
        for k in keys:
            assert torch.equal(a[k], b[k]), (it, k)
    assert torch.equal(pf0, pf1) and torch.equal(vf0, vf1) and torch.equal(tail0, tail1)
    counts = col._prefetcher.transport_counts
    assert counts["carried"] >= 1, counts                              # the default transport of the headline
    assert noise.STATS["parallel_blocks"] - par0 >= 3                  # every block came from the multi-threaded draw
    assert col._prefetcher.dropped_blocks == 0


def test_cfg2_iteration_vs_oracle(errlog):
    """The headline configuration against the CPU oracle, directly: 2048 envs x 128 steps with the reference's noise stream
    (torchrl/collector/on_policy.py:90-155), GAE (on_rl_algo.py:22-33, replay_buffers/on_policy.py:16-44) and one pass of
    four B = 65 536 updates (ppo.py:124-152) -- the 128-workgroup rollout grid, the 256-workgroup gradient grid and the ring
    offsets at N = 2048 tied to the oracle rather than to themselves.  Tolerances of SURVEY.md section 8 (a1 / a6 / a11):
    buffers abs 1e-5, advantages abs 2e-5 / rel 1e-3, info scalars rel 1e-4 / abs 1e-5, parameters abs 1e-6."""
    from oracle import replay as oreplay
    from oracle.collector import VecOnPolicyCollectorOracle
    from oracle.ppo import PPOOracle
    from oracle.synth_env import SynthVecEnvCPU
    seed = 5
    pf, vf, env, buf, col, agent = make(N, seed=seed, noise="host")
    agent.logger.infos.clear()
    pf_p = [p.detach().cpu().clone() for p in pf._mlp2_param_list()]
    vf_p = [p.detach().cpu().clone() for p in vf._mlp2_param_list()]
    ls = pf.logstd.detach().cpu().clone()
    torch.manual_seed(seed)
    col.train_one_epoch()
    agent.current_epoch = 1
    np.random.seed(seed + 100)
    agent.update_per_epoch()
    torch.cuda.synchronize()
    infos = list(agent.logger.infos)
    assert len(infos) == N * T // B == 4

    oenv = SynthVecEnvCPU(N, horizon=50)
    oenv.seed(seed)
    ring = oreplay.RingOracle(N * T, env_nums=N, time_limit_filter=True)
    ocol = VecOnPolicyCollectorOracle(oenv, ring, pf_p, ls, vf_p, epoch_frames=N * T, max_episode_frames=40)
    torch.manual_seed(seed)
    ocol.train_one_epoch()
    o = PPOOracle(pf_p, ls, vf_p, plr=3e-4, vlr=3e-4, entropy_coeff=0.005, clip_para=0.2, opt_epochs=1,
                  num_epochs=10, batch_size=B, discount=0.99, tau=0.95)
    np.random.seed(seed + 100)
    want_infos = o.epoch(ring, 1)
    assert float(ring.data["terminals"].sum()) > 0                                                  # episodes ended inside the rollout

    for k in ("obs", "next_obs", "acts", "values", "rewards", "terminals", "time_limits"):
        err = np.abs(getattr(buf, "_" + k).cpu().numpy().astype(np.float64) - ring.data[k]).max()
        errlog("cfg 2 full size: buffer %s abs" % k, err, 1e-5)
        assert err < 1e-5, (k, err)
    for k, name in (("advs", "advs"), ("estimate_returns", "estimate_returns")):
        got, want = getattr(buf, "_" + k).cpu().numpy().astype(np.float64), ring.data[name]
        errlog("cfg 2 full size: %s, max of |got - want| / (2e-5 + 1e-3 |want|)" % k,
               (np.abs(got - want) / (2e-5 + 1e-3 * np.abs(want))).max(), 1.0)
        np.testing.assert_allclose(got, want, rtol=1e-3, atol=2e-5)
    keys = sorted(want_infos[0].keys())
    assert sorted(infos[0].keys()) == keys and len(want_infos) == 4
    got = np.array([[i[k] for k in keys] for i in infos])
    want = np.array([[i[k] for k in keys] for i in want_infos])
    rel = np.abs(got - want) / (1e-5 + 1e-4 * np.abs(want))
    errlog("cfg 2 full size: info scalars, max of |got - want| / (1e-5 + 1e-4 |want|)", rel.max(), 1.0)
    bad = np.argwhere(rel > 1.0)
    assert len(bad) == 0, [(int(r), keys[c], float(got[r, c]), float(want[r, c])) for r, c in bad]
    got_p = torch.cat([p.detach().reshape(-1) for p in pf._mlp2_param_list()] + [pf.logstd.detach().reshape(-1)] +
                      [p.detach().reshape(-1) for p in vf._mlp2_param_list()]).cpu()
    want_p = torch.cat([p.detach().reshape(-1) for p in o.pf] + [o.logstd.detach().reshape(-1)] +
                       [p.detach().reshape(-1) for p in o.vf])
    perr = (got_p - want_p).abs().max().item()
    errlog("cfg 2 full size: post-epoch params abs (4 updates of 65 536)", perr, 1e-6)
    assert perr < 1e-6, perr
