"""PPO / A2C for network shapes the fused kernels are not instantiated for (other observation / action sizes,
widths, depths): dense-layer GEMMs + trl_ppo_generic_losses_f32.  The same engine forced onto the benchmark shape
is checked against the REFERENCE's outputs in test_product_gpu.py / test_a2c_gpu.py (engine = generic); here the
shapes differ and the check is against the CPU oracle (itself pinned to those reference outputs)."""
import numpy as np
import pytest
import torch

from oracle import nets
from oracle.ppo import A2COracle, PPOOracle

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


class _Stub:
    epoch_frames = 0


class _Log:
    def __init__(self): self.infos = []
    def add_update_info(self, d): self.infos.append(dict(d))
    def add_epoch_info(self, *a, **k): pass
    def log(self, *a): pass
    def finish(self): pass


def build(D, A, hidden, act_cls, tanh_action, algo_cls, **algo_kw):
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.env.synth import SynthVecEnv
    net = dict(hidden_shapes=hidden, append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=act_cls)
    torch.manual_seed(D * 100 + A)
    pf = policies.GuassianContPolicyBasicBias(input_shape=D, output_shape=A, tanh_action=tanh_action, **net)
    vf = networks.Net(input_shape=(D,), output_shape=1, **net)
    with torch.no_grad():                                              # leave the near-zero init of the heads
        for m in (pf, vf):
            m.seq_append_fcs[-1].weight.mul_(30.0)
        pf.logstd.copy_(torch.linspace(-1.5, -0.5, A))
    flat = lambda m: [p.detach().clone() for l in (list(m.base.seq_fcs) + list(m.seq_append_fcs))
                      if isinstance(l, torch.nn.Linear) for p in (l.weight, l.bias)]
    oracle_params = (flat(pf), pf.logstd.detach().clone(), flat(vf))
    agent = algo_cls(pf=pf, vf=vf, plr=3e-4, vlr=1e-3, tau=0.95, shuffle=True, discount=0.99, num_epochs=10,
                     batch_size=96, gae=True, env=SynthVecEnv(4, obs_dim=D, act_dim=A, device=DEV), replay_buffer=None,
                     collector=_Stub(), logger=_Log(), device=DEV, save_dir=None, **algo_kw)
    return pf, vf, agent, oracle_params


@pytest.mark.parametrize("D,A,hidden,act,tanh_action,clipped", [
    (11, 3, [32, 48, 16], "tanh", True, False),       # Hopper-sized, three uneven hidden layers
    (3, 1, [40], "relu", False, True),                # Pendulum-sized, one hidden layer, no tanh squashing, clipped value loss
    (27, 8, [128, 128], "tanh", True, False),         # wide
])
def test_ppo_update_other_shapes_vs_oracle(D, A, hidden, act, tanh_action, clipped):
    from torchrl.algo import PPO
    act_cls = {"tanh": torch.nn.Tanh, "relu": torch.nn.ReLU}[act]
    pf, vf, agent, (pf0, ls0, vf0) = build(D, A, hidden, act_cls, tanh_action, PPO, clip_para=0.2, opt_epochs=2,
                                          entropy_coeff=0.01, clipped_value_loss=clipped)
    assert type(agent.engine()).__name__ == "_GenericPPO"
    ref = PPOOracle(pf0, ls0, vf0, plr=3e-4, vlr=1e-3, entropy_coeff=0.01, clip_para=0.2, clipped_value_loss=clipped,
                    act=act, tanh_action=tanh_action)
    gen = torch.Generator().manual_seed(7)
    B = 96
    for step in range(3):
        obs = torch.randn(B, D, generator=gen)
        with torch.no_grad():                                          # actions the CURRENT target policy could have produced
            mean = nets.mlp(obs, ref.tpf, act)
            pre = mean + torch.exp(ref.tlogstd) * torch.randn(B, A, generator=gen)
            acts = torch.tanh(pre) if tanh_action else pre
        batch = {"obs": obs.numpy(), "acts": acts.numpy(), "advs": torch.randn(B, 1, generator=gen).numpy() * 2 + 0.3,
                 "values": torch.randn(B, 1, generator=gen).numpy(), "estimate_returns": torch.randn(B, 1, generator=gen).numpy()}
        want = ref.update(batch)
        ref.sync_target()                                              # ppo.py:33: target_pf <- pf once per epoch; here per step
        got = agent.update(batch)
        agent.engine().sync_target_pf()
        assert sorted(got) == sorted(want)
        np.testing.assert_allclose([got[k] for k in sorted(want)], [want[k] for k in sorted(want)], rtol=3e-4, atol=5e-5)
    for mod, params in ((pf, ref.pf), (vf, ref.vf)):
        lin = [l for l in (list(mod.base.seq_fcs) + list(mod.seq_append_fcs)) if isinstance(l, torch.nn.Linear)]
        for k, l in enumerate(lin):
            assert (l.weight.cpu() - params[2 * k].detach()).abs().max().item() < 3e-6
            assert (l.bias.cpu() - params[2 * k + 1].detach()).abs().max().item() < 3e-6
    assert (pf.logstd.cpu() - ref.logstd.detach()).abs().max().item() < 3e-6


def test_a2c_update_other_shape_vs_oracle():
    from torchrl.algo import A2C
    D, A, hidden = 8, 2, [24, 24, 24]
    pf, vf, agent, (pf0, ls0, vf0) = build(D, A, hidden, torch.nn.Tanh, True, A2C, entropy_coeff=0.01)
    assert type(agent.engine()).__name__ == "_GenericPPO"
    ref = A2COracle(pf0, ls0, vf0, plr=3e-4, vlr=1e-3, entropy_coeff=0.01, act="tanh", tanh_action=True)
    gen = torch.Generator().manual_seed(3)
    B = 64
    batch = {"obs": torch.randn(B, D, generator=gen).numpy(), "acts": (torch.rand(B, A, generator=gen) * 1.6 - 0.8).numpy(),
             "advs": torch.randn(B, 1, generator=gen).numpy(), "estimate_returns": torch.randn(B, 1, generator=gen).numpy()}
    for _ in range(2):
        want, got = ref.update(batch), agent.update(batch)
        assert sorted(got) == sorted(want)
        np.testing.assert_allclose([got[k] for k in sorted(want)], [want[k] for k in sorted(want)], rtol=3e-4, atol=5e-5)
    assert (pf.logstd.cpu() - ref.logstd.detach()).abs().max().item() < 3e-6


def test_collect_and_train_on_another_env_shape():
    """Collector (per-step launch sequence on the dense-layer kernels) + GAE + PPO epochs on an 11-obs / 3-act env."""
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.algo import PPO
    from torchrl.collector.on_policy import VecOnPolicyCollector
    from torchrl.env.synth import SynthVecEnv
    from torchrl.replay_buffers.on_policy import OnPolicyReplayBuffer
    N, T, D, A = 32, 16, 11, 3
    net = dict(hidden_shapes=[32, 48], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.Tanh)
    pf = policies.GuassianContPolicyBasicBias(input_shape=D, output_shape=A, tanh_action=True, **net)
    vf = networks.Net(input_shape=(D,), output_shape=1, **net)
    env, eval_env = (SynthVecEnv(N, obs_dim=D, act_dim=A, horizon=12, device=DEV) for _ in range(2))
    env.seed(3)
    buf = OnPolicyReplayBuffer(N * T, env_nums=N, time_limit_filter=True)
    col = VecOnPolicyCollector(vf, env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=DEV, train_render=False,
                               epoch_frames=N * T, max_episode_frames=9, eval_episodes=1, noise_mode="device")
    assert col._spec is None
    logger = _Log()
    agent = PPO(pf=pf, vf=vf, plr=3e-4, vlr=3e-4, clip_para=0.2, opt_epochs=2, tau=0.95, shuffle=True, entropy_coeff=0.005,
                discount=0.99, num_epochs=10, batch_size=N * 4, gae=True, env=env, replay_buffer=buf, collector=col,
                logger=logger, device=DEV, save_dir=None)
    p0 = torch.cat([p.detach().reshape(-1) for p in pf.parameters()]).clone()
    for epoch in range(2):
        res = col.train_one_epoch()
        assert np.isfinite(res["train_epoch_reward"])
        agent.current_epoch = epoch
        agent.update_per_epoch()
    assert len(logger.infos) == 2 * 2 * (T // 4)
    assert all(np.isfinite(list(i.values())).all() for i in logger.infos)
    # first minibatch of an epoch: log pi == log pi_old (same kernels wrote old_logp) -> ratio exactly 1
    assert logger.infos[0]["ratio/max"] == 1.0 and logger.infos[0]["ratio/min"] == 1.0
    assert (torch.cat([p.detach().reshape(-1) for p in pf.parameters()]) - p0).abs().max() > 0
    ev = col.eval_one_epoch()
    assert len(ev["eval_rewards"]) == N and ev["eval_traj_length"] == 12


def test_graph_replayed_per_step_rollout_equals_eager(monkeypatch):
    """Shapes without a persistent rollout kernel collect through per-step launch sequences; from the third rollout on
    the whole sequence replays as one HIP graph.  Same launches and the same up-front noise draw: identical buffers."""
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.collector.on_policy import VecOnPolicyCollector
    from torchrl.env.synth import SynthVecEnv
    from torchrl.replay_buffers.on_policy import OnPolicyReplayBuffer
    N, T, D, A = 16, 12, 9, 2
    results = []
    for no_graph in ("1", "0"):
        monkeypatch.setenv("TRL_NO_GRAPH", no_graph)
        torch.manual_seed(0)
        net = dict(hidden_shapes=[24, 40], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.Tanh)
        pf = policies.GuassianContPolicyBasicBias(input_shape=D, output_shape=A, tanh_action=True, **net)
        vf = networks.Net(input_shape=(D,), output_shape=1, **net)
        env, eval_env = (SynthVecEnv(N, obs_dim=D, act_dim=A, horizon=7, device=DEV) for _ in range(2))
        env.seed(1)
        buf = OnPolicyReplayBuffer(N * T, env_nums=N, time_limit_filter=True)
        col = VecOnPolicyCollector(vf, env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=DEV, train_render=False,
                                   epoch_frames=N * T, max_episode_frames=5, eval_episodes=1, noise_mode="device")
        snaps = []
        for _ in range(4):
            res = col.train_one_epoch()
            snaps.append({k: getattr(buf, "_" + k).clone() for k in ("obs", "next_obs", "acts", "values", "rewards",
                                                                     "terminals", "old_logp")}
                         | {"reward": res["train_epoch_reward"], "n_eps": len(res["train_rewards"])})
        assert (col._roll_graph["graph"] is not None) == (no_graph == "0")
        results.append(snaps)
    for a, b in zip(*results):
        assert a["reward"] == b["reward"] and a["n_eps"] == b["n_eps"]
        for k in a:
            if isinstance(a[k], torch.Tensor):
                assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("D,A,act,tanh_action,clipped,B", [
    (11, 3, "tanh", True, False, 96),        # Hopper-sized: the runtime-dims instantiation of the fused kernels
    (8, 2, "relu", True, True, 100),         # Swimmer-sized, clipped value loss, a batch that is no multiple of 16
    (4, 1, "tanh", False, False, 96),        # a single action, no tanh squashing
    (17, 8, "tanh", True, False, 112),       # the full tile: 17 inputs, 8 actions
    (16, 6, "relu", True, False, 96),        # no 17th feature
    (2, 5, "tanh", True, True, 100),
    (27, 8, "tanh", True, False, 112),       # Ant-sized: the wide tile (a second 16-feature group on the matrix pipe)
    (18, 2, "relu", True, True, 100),        # one feature into the second group, ragged batch
    (32, 6, "tanh", False, False, 96),       # the full wide tile
])
def test_fused_ppo_update_at_other_input_and_action_sizes_vs_oracle(D, A, act, tanh_action, clipped, B, errlog):
    """64-wide two-layer networks with any D in [2, 32], A in [1, 8] stay on the fused two-launch update
    (trl_ppo_minibatch_grad_f32's runtime-dims instantiation + trl_ppo_reduce_adam_f32): three chained updates against
    the CPU oracle at the contract of the benchmark shape (scalars rel 1e-4 / abs 1e-5, post-step parameters abs 1e-6)."""
    from torchrl.algo import PPO
    act_cls = {"tanh": torch.nn.Tanh, "relu": torch.nn.ReLU}[act]
    pf, vf, agent, (pf0, ls0, vf0) = build(D, A, [64, 64], act_cls, tanh_action, PPO, clip_para=0.2, opt_epochs=2,
                                          entropy_coeff=0.01, clipped_value_loss=clipped)
    assert type(agent.engine()).__name__ == "_FusedPPO"
    ref = PPOOracle(pf0, ls0, vf0, plr=3e-4, vlr=1e-3, entropy_coeff=0.01, clip_para=0.2, clipped_value_loss=clipped,
                    act=act, tanh_action=tanh_action)
    gen = torch.Generator().manual_seed(11 * D + A)
    for step in range(3):
        obs = torch.randn(B, D, generator=gen)
        with torch.no_grad():
            mean = nets.mlp(obs, ref.tpf, act)
            pre = mean + torch.exp(ref.tlogstd) * torch.randn(B, A, generator=gen)
            acts = torch.tanh(pre) if tanh_action else pre
        batch = {"obs": obs.numpy(), "acts": acts.numpy(), "advs": torch.randn(B, 1, generator=gen).numpy() * 2 + 0.3,
                 "values": torch.randn(B, 1, generator=gen).numpy(), "estimate_returns": torch.randn(B, 1, generator=gen).numpy()}
        want = ref.update(batch)
        ref.sync_target()
        got = agent.update(batch)
        agent.engine().sync_target_pf()
        assert sorted(got) == sorted(want)
        keys = sorted(want)
        g, w = np.array([got[k] for k in keys]), np.array([want[k] for k in keys])
        # One action dimension: the unbiased std over ONE element is NaN in the reference too (ppo.py:82-85: `.std()` of a
        # 1-element tensor) -- exactly these keys, in both, and nothing else; every other scalar is compared by value.
        nan_keys = {"log_std/std", "std/std"} & set(keys) if A == 1 else set()
        assert {k for k, v in zip(keys, w) if np.isnan(v)} == nan_keys, "oracle NaNs"
        assert {k for k, v in zip(keys, g) if np.isnan(v)} == nan_keys, "kernel NaNs"
        fin = np.array([k not in nan_keys for k in keys])
        assert fin.sum() >= len(keys) - 2 and np.isfinite(g[fin]).all()
        errlog("step %d info scalars: max |got - want| / (1e-5 + 1e-4 |want|)" % step,
               (np.abs(g[fin] - w[fin]) / (1e-5 + 1e-4 * np.abs(w[fin]))).max(), 1.0)
        np.testing.assert_allclose(g[fin], w[fin], rtol=1e-4, atol=1e-5)
    perr = 0.0
    for mod, params in ((pf, ref.pf), (vf, ref.vf)):
        lin = [l for l in (list(mod.base.seq_fcs) + list(mod.seq_append_fcs)) if isinstance(l, torch.nn.Linear)]
        for k, l in enumerate(lin):
            perr = max(perr, (l.weight.cpu() - params[2 * k].detach()).abs().max().item(),
                       (l.bias.cpu() - params[2 * k + 1].detach()).abs().max().item())
    perr = max(perr, (pf.logstd.cpu() - ref.logstd.detach()).abs().max().item())
    errlog("post-step params abs (3 updates)", perr, 1e-6)
    assert perr < 1e-6, perr


@pytest.mark.parametrize("D,A", [(11, 3), (27, 8)])
def test_collect_and_train_with_the_fused_update_on_a_hopper_sized_env(D, A):
    """11 observations / 3 actions (Hopper) and 27 / 8 (Ant: the wide tile), 64-wide networks: ONE rollout launch per
    epoch (runtime-dims instantiations of the persistent kernel) feeds the fused update kernels -- 2 launches per
    minibatch; log pi_old comes from the rollout kernel, log pi from the update kernel: the first ratio is 1 up to round-off."""
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.algo import PPO
    from torchrl.collector.on_policy import VecOnPolicyCollector
    from torchrl.env.synth import SynthVecEnv
    from torchrl.replay_buffers.on_policy import OnPolicyReplayBuffer
    N, T = 32, 16
    net = dict(hidden_shapes=[64, 64], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.Tanh)
    torch.manual_seed(1)
    pf = policies.GuassianContPolicyBasicBias(input_shape=D, output_shape=A, tanh_action=True, **net)
    vf = networks.Net(input_shape=(D,), output_shape=1, **net)
    env, eval_env = (SynthVecEnv(N, obs_dim=D, act_dim=A, horizon=12, device=DEV) for _ in range(2))
    env.seed(3)
    buf = OnPolicyReplayBuffer(N * T, env_nums=N, time_limit_filter=True)
    col = VecOnPolicyCollector(vf, env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=DEV, train_render=False,
                               epoch_frames=N * T, max_episode_frames=9, eval_episodes=1, noise_mode="device")
    assert col._mlp2 is None and col._spec == (D, 64, A, 0)             # the runtime-dims rollout kernel (round 4) ...
    logger = _Log()
    agent = PPO(pf=pf, vf=vf, plr=3e-4, vlr=3e-4, clip_para=0.2, opt_epochs=2, tau=0.95, shuffle=True, entropy_coeff=0.005,
                discount=0.99, num_epochs=10, batch_size=N * 4, gae=True, env=env, replay_buffer=buf, collector=col,
                logger=logger, device=DEV, save_dir=None)
    assert type(agent.engine()).__name__ == "_FusedPPO"                 # ... and the fused update
    p0 = pf.flat_params().clone()
    for epoch in range(3):                                              # eager, captured, replayed
        res = col.train_one_epoch()
        assert np.isfinite(res["train_epoch_reward"])
        agent.current_epoch = epoch
        agent.update_per_epoch()
    torch.cuda.synchronize()
    assert len(logger.infos) == 3 * 2 * (T // 4)
    assert all(np.isfinite(list(i.values())).all() for i in logger.infos)
    assert abs(logger.infos[0]["ratio/max"] - 1.0) < 1e-5 and abs(logger.infos[0]["ratio/min"] - 1.0) < 1e-5
    assert (pf.flat_params() - p0).abs().max() > 0
    ev = col.eval_one_epoch()
    assert len(ev["eval_rewards"]) == N and ev["eval_traj_length"] == 12


@pytest.mark.parametrize("D,A,act,tanh_action", [(11, 3, "tanh", True), (27, 8, "tanh", True), (8, 2, "relu", True),
                                                 (17, 6, "tanh", False), (18, 1, "tanh", True), (32, 8, "relu", True),
                                                 (2, 1, "tanh", True), (16, 4, "tanh", True)])
def test_runtime_dims_rollout_kernel_matches_the_oracle_collector(D, A, act, tanh_action, errlog, monkeypatch):
    """the persistent rollout kernel for other task shapes (Hopper 11 / 3, Ant 27 / 8, Swimmer 8 / 2,
    the edges of both tiles) -- ONE launch for the whole epoch -- against VecOnPolicyCollectorOracle
    (torchrl/collector/on_policy.py:90-155 restated) on a synthetic env of that shape, in the reference's noise stream:
    all 7 buffers + log pi_old, the epoch reward and the finished-episode list, with env time-limit resets and the
    collector's over-length bootstrap both firing.  (D = 17 / A = 6 is the compile-time instantiation, here without tanh
    squashing; D = 16, 17, 18 and 32 are the edges of the 17- and 32-feature tiles.)"""
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.collector.on_policy import VecOnPolicyCollector
    from torchrl.env.synth import SynthVecEnv
    from torchrl.replay_buffers.on_policy import OnPolicyReplayBuffer
    from torchrl_amd import _C
    from oracle import replay as oreplay
    from oracle.collector import VecOnPolicyCollectorOracle
    from oracle.synth_env import SynthVecEnvCPU
    N, T, horizon, max_frames, seed = 48, 24, 10, 7, 5
    act_cls = {"tanh": torch.nn.Tanh, "relu": torch.nn.ReLU}[act]
    net = dict(hidden_shapes=[64, 64], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=act_cls)
    torch.manual_seed(D * 31 + A)
    pf = policies.GuassianContPolicyBasicBias(input_shape=D, output_shape=A, tanh_action=tanh_action, **net)
    vf = networks.Net(input_shape=(D,), output_shape=1, **net)
    with torch.no_grad():
        for m in (pf, vf):
            m.seq_append_fcs[-1].weight.mul_(30.0)
        pf.logstd.copy_(torch.linspace(-1.5, -0.5, A))
    flat = lambda m: [p.detach().clone() for l in (list(m.base.seq_fcs) + list(m.seq_append_fcs))
                      if isinstance(l, torch.nn.Linear) for p in (l.weight, l.bias)]
    pf0, ls0, vf0 = flat(pf), pf.logstd.detach().clone(), flat(vf)
    env, eval_env = (SynthVecEnv(N, obs_dim=D, act_dim=A, horizon=horizon, device=DEV) for _ in range(2))
    env.seed(seed)
    buf = OnPolicyReplayBuffer(N * T, env_nums=N, time_limit_filter=True)
    col = VecOnPolicyCollector(vf, env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=DEV, train_render=False,
                               epoch_frames=N * T, max_episode_frames=max_frames, eval_episodes=1, noise_mode="host")
    assert col._spec == (D, 64, A, {"tanh": 0, "relu": 1}[act])
    launches = []
    real = _C.rollout
    monkeypatch.setattr(_C, "rollout", lambda a, dev: (launches.append(int(a.n_steps)), real(a, dev))[1])
    torch.manual_seed(seed)
    res = col.train_one_epoch()
    got_reward, got_eps = float(res["train_epoch_reward"]), sorted(float(x) for x in res["train_rewards"])
    assert launches == [T]                                             # the whole epoch was ONE rollout call
    oenv = SynthVecEnvCPU(N, horizon=horizon, obs_dim=D, act_dim=A)
    oenv.seed(seed)
    ring = oreplay.RingOracle(N * T, env_nums=N, time_limit_filter=True)
    ocol = VecOnPolicyCollectorOracle(oenv, ring, pf0, ls0, vf0, epoch_frames=N * T, max_episode_frames=max_frames,
                                      act=act, tanh_action=tanh_action)
    torch.manual_seed(seed)
    ores = ocol.train_one_epoch()
    for k in ("obs", "next_obs", "acts", "values", "rewards", "terminals", "time_limits"):
        a, b = getattr(buf, "_" + k).cpu().numpy().astype(np.float64), np.asarray(ring.data[k], dtype=np.float64).reshape(T, N, -1)
        err = np.abs(a - b).max()
        errlog("rt rollout %s D=%d A=%d abs" % (k, D, A), err, 1e-5)
        assert err < 1e-5, (k, err)
    assert float(buf._terminals.sum()) > 0 and float((buf._terminals - buf._time_limits).abs().sum()) > 0   # both reset kinds
    assert abs(got_reward - float(ores["train_epoch_reward"])) < 1e-3 * max(1.0, abs(got_reward))
    np.testing.assert_allclose(got_eps, sorted(float(x) for x in ores["train_rewards"]), atol=1e-4)
    # log pi_old of the stored actions under the collecting policy (what target_pf recomputes, ppo.py:54-56)
    from oracle import nets as onets
    obs_t = torch.as_tensor(ring.data["obs"], dtype=torch.float32).reshape(T * N, D)
    act_t = torch.as_tensor(ring.data["acts"], dtype=torch.float32).reshape(T * N, A)
    with torch.no_grad():
        want_lp = onets.policy_update_terms(obs_t, act_t, pf0, ls0, act, tanh_action)["log_prob"].reshape(T, N, 1).numpy()
    np.testing.assert_allclose(buf._old_logp.cpu().numpy(), want_lp, rtol=2e-4, atol=2e-4)
