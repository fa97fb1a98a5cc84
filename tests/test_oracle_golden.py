"""Pin the CPU oracle against fixtures produced by the imported reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import nets, replay
from oracle.collector import VecOnPolicyCollectorOracle
from oracle.ppo import PPOOracle
from oracle.synth_env import SynthVecEnvCPU


def regen_gae_inputs(args):
    T, N, seed, p_term, p_tl = int(args[0]), int(args[1]), int(args[2]), args[3], args[4]
    rs = np.random.RandomState(seed)
    r = rs.randn(T, N, 1).astype(np.float32)
    v = rs.randn(T, N, 1).astype(np.float32)
    d = rs.rand(T, N, 1) < p_term
    tl = (rs.rand(T, N, 1) < p_tl) & d
    lv = rs.randn(N, 1).astype(np.float32)
    return r, v, d, tl, lv


@pytest.mark.parametrize("tag", ["kat", "small", "ragged", "one", "cfg2"])
def test_gae_and_discount_match_reference(golden, tag):
    g = golden("gae")
    if tag == "cfg2":
        r, v, d, tl, lv = regen_gae_inputs(g["cfg2_args"])
        gamma, tau, stride = g["cfg2_args"][5], g["cfg2_args"][6], int(g["cfg2_args"][7])
    else:
        r, v, d, tl, lv = (g[f"{tag}_{k}"] for k in
                           ("rewards", "values", "terminals", "time_limits", "last_value"))
        gamma, tau, stride = (0.99, 0.95, 1) if tag == "kat" else (g[f"{tag}_args"][5], g[f"{tag}_args"][6], 1)
    for filt in (0, 1):
        adv, ret = replay.gae(r, v, d, tl, lv, gamma, tau, bool(filt))
        np.testing.assert_allclose(adv[:, ::stride], g[f"{tag}_gae{filt}_advs"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(ret[:, ::stride], g[f"{tag}_gae{filt}_rets"], rtol=0, atol=1e-12)
        adv, ret = replay.discounted_return(r, v, d, tl, lv, gamma, bool(filt))
        np.testing.assert_allclose(adv[:, ::stride], g[f"{tag}_disc{filt}_advs"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(ret[:, ::stride], g[f"{tag}_disc{filt}_rets"], rtol=0, atol=1e-12)


def test_gae_kat_values_from_survey(golden):
    g = golden("gae")
    want = np.array([[0.799128, 1.24545], [-0.103, 0.9], [0.0, 0.214878], [0.899, 1.397]])
    np.testing.assert_allclose(g["kat_gae1_advs"][..., 0], want, atol=1e-6)


def test_index_streams_bit_exact(golden):
    g = golden("index_streams")
    np.random.seed(0)
    assert np.array_equal(np.random.permutation(8), g["perm8_seed0"])
    assert list(g["perm8_seed0"]) == [6, 2, 1, 7, 3, 0, 5, 4]
    np.random.seed(0)
    assert np.array_equal(np.random.randint(0, 100, 4), g["randint100x4_seed0"])
    assert list(g["randint100x4_seed0"]) == [44, 47, 64, 67]


def test_epoch_minibatches_match_reference(golden):
    g = golden("index_streams")
    T, N, B, E, seed = (int(x) for x in g["oi_args"])
    ring = replay.RingOracle(T * N, env_nums=N)
    for t in range(T):
        ring.add({"obs": g["oi_obs"][t], "acts": g["oi_acts"][t], "advs": g["oi_advs"][t]})
    np.random.seed(seed)
    got = []
    for _ in range(E):
        for _idx, b in ring.epoch_minibatches(B, ["obs", "acts", "advs"], True):
            got.append(np.concatenate([b["obs"], b["acts"], b["advs"]], -1))
    assert np.array_equal(np.stack(got), g["oi_batches"])
    got = [np.concatenate([b["obs"], b["acts"], b["advs"]], -1)
           for _i, b in ring.epoch_minibatches(B, ["obs", "acts", "advs"], False)]
    assert np.array_equal(np.stack(got), g["oi_batches_noshuffle"])


def test_ring_and_uniform_sample_match_reference(golden):
    g = golden("index_streams")
    size, N, B, seed = (int(x) for x in g["ring_args"])
    ring = replay.RingOracle(size, env_nums=N)
    np.random.seed(seed)
    for t in range(7):
        ring.add({"obs": g["ring_adds"][t], "rewards": g["ring_rew"][t]})
        assert ring.size == g["ring_sizes"][t] and ring.top == g["ring_tops"][t]
        _idx, b = ring.sample_rows(B, ["obs", "rewards"])
        assert np.array_equal(np.concatenate([b["obs"], b["rewards"]], -1), g["ring_batches"][t])
    assert np.array_equal(ring.data["obs"], g["ring_obs"])
    assert np.array_equal(ring.data["rewards"], g["ring_rewards"])
    with pytest.raises(AssertionError):
        ring.sample_rows(B + 1, ["obs"])


def params_from(g, prefix, with_logstd):
    names = sorted(k for k in g.files if k.startswith(prefix))
    base = [k for k in names if "base__seq_fcs" in k]
    head = [k for k in names if "seq_append_fcs" in k]
    order = sorted(base, key=lambda k: (int(k.split("__")[-2]), "bias" in k)) + \
        sorted(head, key=lambda k: "bias" in k)
    ps = [torch.tensor(g[k]) for k in order]
    ls = torch.tensor(g[prefix + "logstd"]) if with_logstd else None
    return ps, ls


def test_init_distribution_matches_reference(golden):
    g = golden("net_init")
    pf, ls = params_from(g, "pf_", True)
    assert [tuple(p.shape) for p in pf] == [(64, 17), (64,), (64, 64), (64,), (6, 64), (6,)]
    assert np.allclose(ls.numpy(), np.log(0.125))
    mine = nets.init_mlp(17, [64, 64], 6)
    for a, b in zip(mine, pf):
        assert a.shape == b.shape
    # hidden bound sqrt(1/out_features) (init.py:8), bias 0.1; head +-3e-3
    assert pf[0].abs().max() <= 1 / 8 and pf[0].abs().max() > 0.11
    assert torch.all(pf[1] == 0.1) and pf[4].abs().max() <= 3e-3
    assert mine[0].abs().max() <= 1 / 8 and torch.all(mine[3] == 0.1) and mine[5].abs().max() <= 3e-3


@pytest.mark.parametrize("tag", ["small", "clipv", "mid"])
def test_ppo_update_matches_reference(golden, tag):
    g = golden("ppo_update")
    B, H, clipv, steps = (int(x) for x in g[f"{tag}_args"])
    pf, ls = params_from(g, f"{tag}_pf0_", True)
    vf, _ = params_from(g, f"{tag}_vf0_", False)
    tpf, tls = params_from(g, f"{tag}_tpf0_", True)
    o = PPOOracle(pf, ls, vf, plr=3e-4, vlr=3e-4, entropy_coeff=0.005, clip_para=0.2,
                  clipped_value_loss=bool(clipv), num_epochs=10)
    o.tpf, o.tlogstd = tpf, tls
    batch = {k: g[f"{tag}_batch_{k}"] for k in ("obs", "acts", "advs", "values", "estimate_returns")}
    for s in range(steps):
        info = o.update(batch)
        keys = [str(k) for k in g[f"{tag}_info{s}_keys"]]
        assert sorted(info.keys()) == keys
        got = np.array([info[k] for k in keys])
        np.testing.assert_allclose(got, g[f"{tag}_info{s}_vals"], rtol=2e-5, atol=2e-6)
        want_pf, want_ls = params_from(g, f"{tag}_pf{s + 1}_", True)
        want_vf, _ = params_from(g, f"{tag}_vf{s + 1}_", False)
        for a, b in zip(o.pf + [o.logstd] + o.vf, want_pf + [want_ls] + want_vf):
            np.testing.assert_allclose(a.detach().numpy(), b.numpy(), rtol=0, atol=1e-6)


@pytest.mark.parametrize("tag", ["small", "surpass", "mixed"])
def test_collect_then_epoch_matches_reference(golden, tag):
    g = golden("collect_epoch")
    N, T, horizon, max_frames, B, seed = (int(x) for x in g[f"{tag}_args"])
    pf, ls = params_from(g, f"{tag}_pf0_", True)
    vf, _ = params_from(g, f"{tag}_vf0_", False)
    env = SynthVecEnvCPU(N, horizon=horizon)
    env.seed(seed)
    ring = replay.RingOracle(N * T, env_nums=N, time_limit_filter=True)
    col = VecOnPolicyCollectorOracle(env, ring, pf, ls, vf, epoch_frames=N * T,
                                     max_episode_frames=max_frames)
    res = col.train_one_epoch(noise=torch.tensor(g[f"{tag}_noise"]))
    for k in ("obs", "next_obs", "acts", "values", "rewards", "terminals", "time_limits"):
        np.testing.assert_allclose(ring.data[k], g[f"{tag}_buf_{k}"], rtol=0, atol=2e-6, err_msg=k)
    assert g[f"{tag}_buf_terminals"].sum() > 0
    np.testing.assert_allclose(res["train_epoch_reward"], g[f"{tag}_train_epoch_reward"], atol=1e-4)
    np.testing.assert_allclose(np.array(res["train_rewards"], dtype=np.float64).reshape(-1),
                               g[f"{tag}_train_rewards"], atol=1e-5)
    np.testing.assert_allclose(col.current_ob, g[f"{tag}_current_ob"], atol=2e-6)

    o = PPOOracle(pf, ls, vf, plr=3e-4, vlr=3e-4, entropy_coeff=0.005, clip_para=0.2,
                  opt_epochs=2, num_epochs=10, batch_size=B, discount=0.99, tau=0.95)
    np.random.seed(seed + 100)
    infos = o.epoch(ring, current_epoch=1)
    np.testing.assert_allclose(ring.data["advs"], g[f"{tag}_advs"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(ring.data["estimate_returns"], g[f"{tag}_rets"], rtol=0, atol=2e-5)
    keys = [str(k) for k in g[f"{tag}_info_keys"]]
    got = np.array([[i[k] for k in keys] for i in infos])
    np.testing.assert_allclose(got, g[f"{tag}_infos"], rtol=1e-4, atol=1e-5)
    want_pf, want_ls = params_from(g, f"{tag}_pf1_", True)
    want_vf, _ = params_from(g, f"{tag}_vf1_", False)
    for a, b in zip(o.pf + [o.logstd] + o.vf, want_pf + [want_ls] + want_vf):
        np.testing.assert_allclose(a.detach().numpy(), b.numpy(), rtol=0, atol=2e-6)


def sac_params(g, prefix):
    names = sorted(k for k in g.files if k.startswith(prefix))
    base = [k for k in names if "base__seq_fcs" in k]
    head = [k for k in names if "seq_append_fcs" in k]
    order = sorted(base, key=lambda k: (int(k.split("__")[-2]), "bias" in k)) + sorted(head, key=lambda k: "bias" in k)
    return [torch.tensor(g[k]) for k in order]


@pytest.mark.parametrize("tag", ["env_limit", "collector_limit", "wrap"])
def test_offpolicy_collector_matches_reference(golden, tag):
    """VecCollector.train_one_epoch (collector/base.py:176-230) as run by the reference: ring contents (including a ring
    that wraps), reset bookkeeping and logged episode returns."""
    from oracle.collector import VecCollectorOracle
    g = golden("collect_offpolicy")
    N, steps, rows, horizon, max_frames, seed = (int(x) for x in g[f"{tag}_args"])
    pf = sac_params(g, f"{tag}_pf_")
    env = SynthVecEnvCPU(N, horizon=horizon)
    env.seed(seed)
    ring = replay.RingOracle(N * rows, env_nums=N)
    col = VecCollectorOracle(env, ring, pf, epoch_frames=N * steps, max_episode_frames=max_frames, act="relu",
                             tanh_action=True)
    res = col.train_one_epoch(noise=torch.tensor(g[f"{tag}_noise"]))
    for k in ("obs", "next_obs", "acts", "rewards", "terminals", "time_limits"):
        np.testing.assert_allclose(ring.data[k], g[f"{tag}_buf_{k}"], rtol=0, atol=2e-6, err_msg=k)
    assert (ring.top, ring.size) == tuple(int(x) for x in g[f"{tag}_top_size"])
    np.testing.assert_allclose(res["train_epoch_reward"], g[f"{tag}_train_epoch_reward"], atol=1e-4)
    np.testing.assert_allclose(np.array(res["train_rewards"], dtype=np.float64).reshape(-1),
                               g[f"{tag}_train_rewards"], atol=1e-5)
    np.testing.assert_allclose(col.current_ob, g[f"{tag}_current_ob"], atol=2e-6)
    np.testing.assert_array_equal(col.current_step, g[f"{tag}_current_step"])


def test_process_parallel_env_of_the_cpu_baseline_matches_reference(golden):
    """oracle.subproc_env.SubProcVecEnvCPU -- the env side of bench.py's cpu_baseline -- against the reference's
    SubProcVecEnv run over the same per-env objects (tests/golden/subproc_vecenv.npz), bit for bit."""
    import functools
    import importlib.util
    import os
    from oracle.subproc_env import SubProcVecEnvCPU
    from oracle.synth_env import SynthSingleEnvCPU
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("_make_golden", os.path.join(here, "golden", "make_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    g = golden("subproc_vecenv")
    N, procs, steps, horizon = (int(x) for x in g["args"])
    env = SubProcVecEnvCPU(procs, N, [functools.partial(SynthSingleEnvCPU, 0, horizon)] * N, SynthSingleEnvCPU(0, horizon))
    try:
        rec = gen.subproc_script(env, N, steps)
    finally:
        env.close()
    assert rec["done"].any() and rec["tl"].any()
    for k, v in rec.items():
        assert v.dtype == g[k].dtype and np.array_equal(v, g[k]), k


def test_eps_greedy_explore_matches_reference(golden):
    from oracle.dqn import EpsGreedyOracle
    g = golden("eps_greedy")
    N, A, D, H, calls, decay = (int(x) for x in g["args"])
    qf = sac_params(g, "qf_")
    pol = EpsGreedyOracle(0.9, 0.15, decay, A)
    np.random.seed(21)
    for c in range(calls):
        with torch.no_grad():
            q = nets.mlp(torch.tensor(g["obs"][c]), qf, "relu").numpy()
        assert np.array_equal(pol.explore(q), g["actions"][c]), c
        assert pol.epsilon == float(g["epsilon"][c])
    assert float(g["epsilon"][-1]) == 0.15 and 0.9 > float(g["epsilon"][0]) > 0.8


def test_eval_loop_matches_reference(golden):
    """VecCollector.eval_one_epoch (collector/base.py:232-280) as run by the reference: the synthetic env with the greedy
    tanh-Gaussian action, and the in-process VecEnv over the pure-Python cart-pole with a greedy Q-network."""
    from oracle.collector import eval_one_epoch
    from torchrl_amd.env.py_envs import CartPoleEnv
    from torchrl_amd.env.vecenv import VecEnv
    g = golden("eval_epoch")
    N, horizon, episodes, seed = (int(x) for x in g["synth_args"])
    pf = sac_params(g, "synth_pf_")
    env = SynthVecEnvCPU(N, horizon=horizon)
    env.seed(seed + 1)
    with torch.no_grad():
        greedy = lambda o: torch.tanh(nets.mlp(torch.as_tensor(o, dtype=torch.float32), pf, "relu")[:, :6]).numpy()
        res = eval_one_epoch(env, greedy, episodes)
    np.testing.assert_allclose(np.array(res["eval_rewards"]).reshape(-1), g["synth_eval_rewards"], atol=2e-5)
    assert res["eval_traj_length"] == float(g["synth_eval_traj_length"])
    N, H, episodes, seed = (int(x) for x in g["cartpole_args"])
    qf = sac_params(g, "cartpole_qf_")
    env = VecEnv(N, CartPoleEnv, ())
    env.seed(seed + 1)
    with torch.no_grad():
        greedy = lambda o: nets.mlp(torch.as_tensor(o, dtype=torch.float32), qf, "relu").max(dim=-1, keepdim=True)[1].numpy()
        res = eval_one_epoch(env, greedy, episodes)
    np.testing.assert_array_equal(np.array(res["eval_rewards"]).reshape(-1), g["cartpole_eval_rewards"])
    assert res["eval_traj_length"] == float(g["cartpole_eval_traj_length"])


@pytest.mark.parametrize("tag", ["h256", "reg"])
def test_twin_sac_q_update_matches_reference(golden, tag):
    from oracle.sac import TwinSACQOracle
    g = golden("twin_sac_q")
    B, H, w_reg, clip, steps = g[f"{tag}_args"]
    o = TwinSACQOracle(sac_params(g, f"{tag}_pf0_"), sac_params(g, f"{tag}_qf10_"), sac_params(g, f"{tag}_qf20_"),
                       plr=3e-4, qlr=1e-3, w_std=w_reg, w_mean=w_reg, grad_clip=clip if clip > 0 else None)
    for s in range(int(steps)):
        batch = {k: g[f"{tag}_s{s}_batch_{k}"] for k in ("obs", "next_obs", "acts", "rewards", "terminals")}
        info = o.update(batch, g[f"{tag}_s{s}_eps1"], g[f"{tag}_s{s}_eps2"])
        keys = [str(k) for k in g[f"{tag}_s{s}_info_keys"]]
        assert sorted(info.keys()) == keys
        got = np.array([info[k] for k in keys])
        np.testing.assert_allclose(got, g[f"{tag}_s{s}_info_vals"], rtol=5e-5, atol=5e-6)
    for name, mine in (("pf", o.pf), ("qf1", o.q1), ("qf2", o.q2), ("tqf1", o.tq1), ("tqf2", o.tq2)):
        for a, b in zip(mine, sac_params(g, f"{tag}_{name}1_")):
            np.testing.assert_allclose(a.detach().numpy(), b.numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(o.log_alpha.detach().numpy(), g[f"{tag}_log_alpha"], atol=1e-7)


def dqn_batches(g, tag):
    """Regenerate the reference's batches: frames come from RandomState(seed) in the generator's call order."""
    B, Q, A, steps, seed = (int(x) for x in g[f"{tag}_args"])
    rs = np.random.RandomState(seed)
    out = []
    for s in range(steps):
        obs = rs.randint(0, 256, size=(B, 4, 84, 84)).astype(np.uint8)
        nobs = rs.randint(0, 256, size=(B, 4, 84, 84)).astype(np.uint8)
        acts = rs.randint(0, A, size=(B, 1) if Q == 1 else (B,))
        rew = rs.randn(B, 1).astype(np.float32)
        term = (rs.rand(B, 1) < 0.2).astype(np.float32)
        assert np.array_equal(acts.astype(np.float32), g[f"{tag}_s{s}_acts"]) and np.array_equal(rew, g[f"{tag}_s{s}_rewards"])
        out.append({"obs": obs, "next_obs": nobs, "acts": acts, "rewards": rew, "terminals": term})
    return out


def dqn_params(g, prefix):
    names = sorted(k for k in g.files if k.startswith(prefix))
    conv = sorted([k for k in names if "seq_convs" in k], key=lambda k: (int(k.split("__")[-2]), "bias" in k))
    fc = sorted([k for k in names if "seq_append_fcs" in k], key=lambda k: (int(k.split("__")[-2]), "bias" in k))
    return [torch.tensor(g[k]) for k in conv + fc]


@pytest.mark.parametrize("tag", ["dqn", "qrdqn"])
def test_dqn_updates_match_reference(golden, tag):
    from oracle.dqn import DQNOracle
    g = golden("dqn")
    B, Q, A, steps, _ = (int(x) for x in g[f"{tag}_args"])
    o = DQNOracle(dqn_params(g, f"{tag}_qf0_"), strides=[4, 2, 1], quantile_num=Q, action_num=A)
    for s, batch in enumerate(dqn_batches(g, tag)):
        info = o.update(batch)
        ref = dict(zip([str(k) for k in g[f"{tag}_s{s}_info_keys"]], g[f"{tag}_s{s}_info_vals"]))
        for k, v in info.items():
            assert abs(v - ref[k]) < 2e-5 * abs(ref[k]) + 2e-6, (k, v, ref[k])
    for mine, name in ((o.q, "qf1"), (o.tq, "tqf1")):
        for a, b in zip(mine, dqn_params(g, f"{tag}_{name}_")):
            np.testing.assert_allclose(a.detach().numpy(), b.numpy(), rtol=0, atol=2e-6)


def test_quantile_huber_matches_reference(golden):
    from oracle.dqn import huber
    g = golden("dqn")
    src = torch.tensor(g["qr_src"], requires_grad=True)
    tgt = torch.tensor(g["qr_tgt"])
    coef = torch.tensor((2 * np.arange(200) + 1) / 400.0, dtype=torch.float32).view(1, -1)
    diff = tgt.unsqueeze(-1) - src.unsqueeze(1)
    loss = (huber(diff) * (coef - (diff.detach() < 0).float()).abs()).mean()
    loss.backward()
    assert abs(loss.item() - float(g["qr_loss"])) < 1e-7
    np.testing.assert_allclose(src.grad.numpy(), g["qr_grad"], atol=1e-9)


# ---------------------------------------------------------------- running observation normaliser
def test_normalizer_oracle_matches_reference_unit(golden):
    from oracle.normalizer import NormalizerOracle
    g = golden("obs_norm")
    nz = NormalizerOracle((17,))
    pos = 0
    for k, n in enumerate(g["unit_sizes"]):
        x = g["unit_x"][pos:pos + n]
        nz.update_estimate(x)
        np.testing.assert_allclose(nz._mean, g["unit_mean"][k], rtol=0, atol=1e-14)
        np.testing.assert_allclose(nz._var, g["unit_var"][k], rtol=1e-13, atol=0)
        assert nz._count == g["unit_count"][k]
        np.testing.assert_allclose(nz.filt(x), g["unit_filt"][pos:pos + n], rtol=1e-13, atol=1e-14)
        pos += n


@pytest.mark.parametrize("tag", ["flow", "flow_surpass"])
def test_normobs_collect_oracle_matches_reference(golden, tag):
    """NormObs(vec env) under the on-policy collector, including the raw-obs-after-partial_reset quirk (Q14)."""
    from oracle.normalizer import NormObsOracle
    g = golden("obs_norm")
    N, T, horizon, max_frames, seed = (int(v) for v in g[tag + "_args"])
    pf, ls = params_from(g, tag + "_pf_", True)
    vf, _ = params_from(g, tag + "_vf_", False)
    env = NormObsOracle(SynthVecEnvCPU(N, horizon=horizon))
    env.seed(seed)
    ring = replay.RingOracle(N * T, env_nums=N, time_limit_filter=True)
    col = VecOnPolicyCollectorOracle(env, ring, pf, ls, vf, epoch_frames=N * T, max_episode_frames=max_frames)
    np.testing.assert_allclose(col.current_ob, g[tag + "_ob0"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(env._obs_normalizer.state(), g[tag + "_state0"], rtol=1e-12, atol=1e-14)
    res = col.train_one_epoch(noise=torch.tensor(g[tag + "_noise"]))
    for k in ("obs", "next_obs", "acts", "values", "rewards", "terminals", "time_limits"):
        np.testing.assert_allclose(ring.data[k], g[tag + "_buf_" + k], rtol=0, atol=3e-6, err_msg=k)
    assert g[tag + "_buf_terminals"].sum() > 0
    np.testing.assert_allclose(env._obs_normalizer.state(), g[tag + "_state1"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(col.current_ob, g[tag + "_current_ob"], atol=3e-6)
    np.testing.assert_allclose(res["train_epoch_reward"], g[tag + "_train_epoch_reward"], atol=1e-4)


@pytest.mark.parametrize("tag", ["small", "mid"])
def test_a2c_update_oracle_matches_reference(golden, tag):
    from oracle.ppo import A2COracle
    g = golden("a2c_update")
    pf, ls = params_from(g, f"{tag}_pf0_", True)
    vf, _ = params_from(g, f"{tag}_vf0_", False)
    o = A2COracle(pf, ls, vf, plr=3e-4, vlr=1e-3, entropy_coeff=0.01)
    batch = {k: g[f"{tag}_batch_{k}"] for k in ("obs", "acts", "advs", "estimate_returns")}
    for s in range(2):
        info = o.update(batch)
        keys = [str(k) for k in g[f"{tag}_info{s}_keys"]]
        assert sorted(info.keys()) == keys
        np.testing.assert_allclose([info[k] for k in keys], g[f"{tag}_info{s}_vals"], rtol=2e-5, atol=2e-6)
        want_pf, want_ls = params_from(g, f"{tag}_pf{s + 1}_", True)
        want_vf, _ = params_from(g, f"{tag}_vf{s + 1}_", False)
        for a, b in zip(o.pf + [o.logstd] + o.vf, want_pf + [want_ls] + want_vf):
            np.testing.assert_allclose(a.detach().numpy(), b.numpy(), rtol=0, atol=1e-6)


@pytest.mark.parametrize("tag", ["ddpg", "ddpg_clip", "td3", "td3_clip"])
def test_ddpg_td3_oracle_matches_reference(golden, tag):
    from oracle.detac import DDPGOracle, TD3Oracle
    g = golden("ddpg_td3")
    B, H, clip, steps = g[tag + "_args"]
    clip = float(clip) or None
    get = lambda name, k: sac_params(g, f"{tag}_{name}{k}_")
    if tag.startswith("ddpg"):
        o = DDPGOracle(get("pf", 0), get("qf1", 0), plr=3e-4, qlr=1e-3, grad_clip=clip)
        pairs = (("pf", o.pf), ("qf1", o.qf), ("tpf", o.tpf), ("tqf1", o.tqf))
    else:
        o = TD3Oracle(get("pf", 0), get("qf1", 0), get("qf2", 0), plr=3e-4, qlr=1e-3, grad_clip=clip)
        pairs = (("pf", o.pf), ("qf1", o.q1), ("qf2", o.q2), ("tpf", o.tpf), ("tqf1", o.tq1), ("tqf2", o.tq2))
    for s in range(int(steps)):
        batch = {k: g[f"{tag}_s{s}_batch_{k}"] for k in ("obs", "next_obs", "acts", "rewards", "terminals")}
        if tag.startswith("ddpg"):
            info = o.update(batch)
        else:
            info = o.update(batch, g[f"{tag}_s{s}_eps_explore"], g[f"{tag}_s{s}_eps_smooth"])
        keys = [str(k) for k in g[f"{tag}_s{s}_info_keys"]]
        assert sorted(info.keys()) == keys, (s, sorted(info.keys()), keys)
        np.testing.assert_allclose([info[k] for k in keys], g[f"{tag}_s{s}_info_vals"], rtol=2e-5, atol=2e-6)
    for name, params in pairs:
        for a, b in zip(params, get(name, 1)):
            np.testing.assert_allclose(a.detach().numpy(), b.numpy(), rtol=0, atol=1e-6)


@pytest.mark.parametrize("tag", ["small", "odd"])
def test_vmpo_oracle_matches_reference(golden, tag):
    """oracle/vmpo.py against what the REFERENCE's VMPO.update produced (tests/golden/vmpo_update.npz)."""
    import torch
    from oracle.vmpo import VMPOOracle
    g = golden("vmpo_update")
    B, D, A, H = (int(v) for v in g[tag + "_args"])
    def flat(prefix):
        sd = {k[len(prefix):]: torch.tensor(g[k]) for k in g.files if k.startswith(prefix)}
        lin = sorted({k.rsplit("__", 1)[0] for k in sd if k.endswith("__weight")},
                     key=lambda n: (0 if n.startswith("base") else 1, n))
        return [sd[n + "__" + w] for n in lin for w in ("weight", "bias")], sd
    pf, sd = flat(tag + "_pf0_")
    vf, _ = flat(tag + "_vf0_")
    ref = VMPOOracle(pf, sd["logstd"], vf, plr=1e-3, vlr=1e-3, eta_eps=0.02, alpha_eps=0.1)
    for s in range(3):
        batch = {k: g[f"{tag}_s{s}_batch_{k}"] for k in ("obs", "acts", "advs", "values", "estimate_returns")}
        info = ref.update(batch)
        keys = [str(k) for k in g[f"{tag}_s{s}_info_keys"]]
        assert sorted(info) == keys
        np.testing.assert_allclose([info[k] for k in keys], g[f"{tag}_s{s}_info_vals"], rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose([ref.eta.item(), ref.alpha.item()], g[f"{tag}_s{s}_eta_alpha"], rtol=1e-6)
    pf1, sd1 = flat(tag + "_pf1_")
    for a, b in zip(ref.pf, pf1):
        assert (a.detach() - b).abs().max().item() < 2e-6
    assert (ref.logstd.detach() - sd1["logstd"]).abs().max().item() < 2e-6


@pytest.mark.parametrize("tag", ["small", "odd"])
def test_trpo_oracle_matches_reference(golden, tag):
    """oracle/trpo.py against what the REFERENCE's TRPO.update / update_vf produced (tests/golden/trpo_update.npz)."""
    import torch
    from oracle.trpo import TRPOOracle
    g = golden("trpo_update")
    B, D, A, H = (int(v) for v in g[tag + "_args"])
    def flat(prefix):
        sd = {k[len(prefix):]: torch.tensor(g[k]) for k in g.files if k.startswith(prefix)}
        lin = sorted({k.rsplit("__", 1)[0] for k in sd if k.endswith("__weight")},
                     key=lambda n: (0 if n.startswith("base") else 1, n))
        return [sd[n + "__" + w] for n in lin for w in ("weight", "bias")], sd
    pf, sd = flat(tag + "_pf0_")
    vf, _ = flat(tag + "_vf0_")
    ref = TRPOOracle(pf, sd["logstd"], vf, vlr=1e-3)
    for s in range(2):
        batch = {k: g[f"{tag}_s{s}_batch_{k}"] for k in ("obs", "acts", "advs", "estimate_returns")}
        info = ref.update(batch)
        keys = [str(k) for k in g[f"{tag}_s{s}_info_keys"]]
        assert sorted(info) == keys
        np.testing.assert_allclose([info[k] for k in keys], g[f"{tag}_s{s}_info_vals"], rtol=1e-4, atol=2e-6)
        # Ten fp32 CG iterations never converge (residual_tol = 1e-10) and the iterate is ill-conditioned: perturbing the
        # initial weights by 1e-7 (relative) moves the first step by 1.2e-5 and the second by 9e-4 (step sizes ~0.04), so
        # that is the agreement any two implementations can have -- the reference against itself on another BLAS included.
        tol = (3e-5, 3e-3)[s]
        pf1, sd1 = flat(f"{tag}_pf{s + 1}_")
        for a, b in zip(ref.pf, pf1):
            assert (a.detach() - b).abs().max().item() < tol
        assert (ref.logstd.detach() - sd1["logstd"]).abs().max().item() < tol
        vinfo = ref.update_vf(batch)
        np.testing.assert_allclose([vinfo[k] for k in sorted(vinfo)], g[f"{tag}_s{s}_vinfo_vals"], rtol=1e-4, atol=2e-6)


FRAME_TAGS = ["done", "surpass", "mixed", "single", "single84"]


def frame_oracle_from_golden(g, tag):
    from oracle.frames import VecFrameRingOracle
    N, rows, steps, H, W, horizon, max_frames, seed = (int(v) for v in g[f"{tag}_args"])
    ring = VecFrameRingOracle(rows, N, k=4)
    ring.collect(g[f"{tag}_frames"], g[f"{tag}_acts"], horizon, max_frames)
    return ring


@pytest.mark.parametrize("tag", FRAME_TAGS)
def test_frame_stack_ring_matches_reference(golden, tag):
    """oracle/frames.py (FrameStack + LazyFrames + MemoryEfficientReplayBuffer restated) against what the reference's
    classes stored and re-encoded: every replay row of every env, stacks byte for byte."""
    g = golden("frame_dedup")
    ring = frame_oracle_from_golden(g, tag)
    rows = int(g[f"{tag}_args"][1])
    assert [ring.bufs[0]._top, ring.bufs[0]._size] == list(g[f"{tag}_top_size"])
    for key in ("obs", "next_obs", "acts", "rewards", "terminals"):
        got = ring.rows_of(key, list(range(rows)))
        assert got.dtype == np.float64                                   # np.array(..., dtype=float)
        assert np.array_equal(got, g[f"{tag}_ref_{key}"].astype(np.float64)), key
    if tag == "single":                                                  # the reference's own random_batch stream
        seed = int(g["single_args"][7])
        np.random.seed(seed + 50)
        for k in range(3):
            idx, batch = ring.bufs[0].random_batch(7, ["obs", "next_obs", "acts", "rewards", "terminals"])
            for key, v in batch.items():
                assert np.array_equal(v, g[f"single_batch{k}_{key}"].astype(np.float64)), (k, key)
        # the N-env sampling rule (B // N rows x all envs) degenerates to the same stream at N == 1
        np.random.seed(seed + 50)
        _, vb = ring.random_batch(7, ["obs"])
        assert np.array_equal(vb["obs"], g["single_batch0_obs"].astype(np.float64))


@pytest.mark.parametrize("tag", ["env_limit", "wrap"])
def test_offpolicy_collector_on_normalised_env_matches_reference(golden, tag):
    """The off-policy collector on a NormObs env (tests/golden/collect_offpolicy_norm.npz): normalised ring rows,
    statistics after the epoch, raw policy input after resets (Q14), greedy evaluation with the copied normaliser."""
    import copy
    from oracle import nets as onets
    from oracle.collector import VecCollectorOracle, eval_one_epoch
    from oracle.normalizer import NormObsOracle
    from oracle.sac import rsample
    g = golden("collect_offpolicy_norm")
    N, steps, rows, horizon, max_frames, seed = (int(x) for x in g[f"{tag}_args"])
    pf = sac_params(g, f"{tag}_pf_")

    class Space:
        shape = (17,)

    def mk(s):
        e = SynthVecEnvCPU(N, horizon=horizon)
        e.observation_space = Space()
        e.seed(s)
        return NormObsOracle(e)
    env, eval_env = mk(seed), mk(seed + 1)
    ring = replay.RingOracle(N * rows, env_nums=N)
    col = VecCollectorOracle(env, ring, pf, epoch_frames=N * steps, max_episode_frames=max_frames, act="relu",
                             tanh_action=True)
    np.testing.assert_allclose(col.current_ob, g[f"{tag}_ob0"], atol=1e-12)
    res = col.train_one_epoch(noise=torch.tensor(g[f"{tag}_noise"]))
    for k in ("obs", "next_obs", "acts", "rewards", "terminals", "time_limits"):
        np.testing.assert_allclose(ring.data[k], g[f"{tag}_buf_{k}"], rtol=0, atol=5e-6, err_msg=k)
    assert (ring.top, ring.size) == tuple(int(x) for x in g[f"{tag}_top_size"])
    np.testing.assert_allclose(env._obs_normalizer.state(), g[f"{tag}_state1"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(col.current_ob, g[f"{tag}_current_ob"], atol=5e-6)
    np.testing.assert_allclose(np.array(res["train_rewards"], dtype=np.float64).reshape(-1), g[f"{tag}_train_rewards"],
                               atol=1e-5)
    eval_env._obs_normalizer = copy.deepcopy(env._obs_normalizer)         # collector/base.py:236-237

    def greedy(obs):
        with torch.no_grad():
            head = onets.mlp(torch.as_tensor(obs, dtype=torch.float32), pf, "relu")
            return rsample(head, torch.zeros(obs.shape[0], 6), True)[0].numpy()
    ev = eval_one_epoch(eval_env, greedy, 1)
    np.testing.assert_allclose(np.array(ev["eval_rewards"], dtype=np.float64).reshape(-1), g[f"{tag}_eval_rewards"],
                               atol=1e-4)
    assert ev["eval_traj_length"] == float(g[f"{tag}_eval_traj_length"])
