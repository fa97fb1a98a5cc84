"""DDPG and TD3 on the HIP path: `update()` against what the REFERENCE's DDPG /
TD3 produced for the same batches, parameters and CPU noise draws (tests/golden/ddpg_td3.npz), plus the
off-policy collector with a fixed-std exploration policy against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _state(g, prefix):
    return {k[len(prefix):].replace("__", "."): torch.tensor(g[k]) for k in g.files if k.startswith(prefix)}


class _Stub:
    epoch_frames = 0


class _Log:
    def add_update_info(self, d): pass
    def add_epoch_info(self, *a, **k): pass
    def log(self, *a): pass
    def finish(self): pass


def _build(tag, clip, B):
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.algo import DDPG, TD3
    from torchrl.env.synth import SynthVecEnv
    D, A, H = 17, 6, 64
    dev = torch.device(DEV)
    net = dict(hidden_shapes=[H, H], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.ReLU)
    pf = policies.FixGuassianContPolicy(input_shape=D, output_shape=A, tanh_action=True, norm_std_explore=0.1, **net)
    qf1 = networks.QNet(input_shape=D + A, output_shape=1, **net)
    qf2 = networks.QNet(input_shape=D + A, output_shape=1, **net)
    kw = dict(env=SynthVecEnv(4, device=dev), replay_buffer=None, collector=_Stub(), logger=_Log(), grad_clip=clip,
              discount=0.99, num_epochs=10, batch_size=B, device=dev, save_dir=None, tau=0.005, use_soft_update=True,
              opt_times=1)
    return pf, qf1, qf2, kw, DDPG, TD3


@pytest.mark.parametrize("tag", ["ddpg", "ddpg_clip", "td3", "td3_clip"])
def test_ddpg_td3_update_matches_reference(golden, tag):
    g = golden("ddpg_td3")
    B, H, clip, steps = g[tag + "_args"]
    pf, qf1, qf2, kw, DDPG, TD3 = _build(tag, float(clip) or None, int(B))
    pf.load_state_dict(_state(g, f"{tag}_pf0_"))
    qf1.load_state_dict(_state(g, f"{tag}_qf10_"))
    if tag.startswith("ddpg"):
        agent = DDPG(pf=pf, qf=qf1, plr=3e-4, qlr=1e-3, **kw)
        mods = (("pf", pf), ("qf1", qf1), ("tpf", agent.target_pf), ("tqf1", agent.target_qf))
    else:
        qf2.load_state_dict(_state(g, f"{tag}_qf20_"))
        agent = TD3(pf=pf, qf1=qf1, qf2=qf2, plr=3e-4, qlr=1e-3, policy_update_delay=2, norm_std_policy=0.2,
                    noise_clip=0.5, **kw)
        mods = (("pf", pf), ("qf1", qf1), ("qf2", qf2), ("tpf", agent.target_pf), ("tqf1", agent.target_qf1),
                ("tqf2", agent.target_qf2))
    for s in range(int(steps)):
        batch = {k: g[f"{tag}_s{s}_batch_{k}"] for k in ("obs", "next_obs", "acts", "rewards", "terminals")}
        torch.manual_seed(200 + s)                                   # TD3: the reference's two CPU draws per update
        info = agent.update(batch)
        keys = [str(k) for k in g[f"{tag}_s{s}_info_keys"]]
        assert sorted(info.keys()) == keys, (s, sorted(info.keys()), keys)
        np.testing.assert_allclose([info[k] for k in keys], g[f"{tag}_s{s}_info_vals"], rtol=2e-4, atol=5e-5)
    for name, mod in mods:
        for k, p in mod.state_dict().items():
            err = np.abs(p.cpu().numpy() - g[f"{tag}_{name}1_{k.replace('.', '__')}"]).max()
            assert err < 3e-6, (name, k, err)
    assert float(agent.pf_optimizer.state[pf.seq_append_fcs[0].weight]["exp_avg"].abs().sum()) > 0


@pytest.mark.parametrize("kind", ["ddpg", "td3"])
def test_graph_replayed_updates_equal_eager_updates(golden, kind, monkeypatch):
    """From the third visit of a configuration on, an update replays a captured HIP graph (per-network Adam step
    counts live on the device; TD3 alternates between two graphs, with and without the delayed policy step):
    identical launches, so parameters and logged statistics equal the eager sequence bit for bit."""
    g = golden("ddpg_td3")
    B = 256
    gen = torch.Generator().manual_seed(5)
    batches = [{"obs": torch.randn(B, 17, generator=gen), "next_obs": torch.randn(B, 17, generator=gen),
                "acts": torch.rand(B, 6, generator=gen) * 2 - 1, "rewards": torch.randn(B, 1, generator=gen),
                "terminals": (torch.rand(B, 1, generator=gen) < 0.1).float()} for _ in range(8)]
    results = []
    for no_graph in ("1", "0"):
        monkeypatch.setenv("TRL_NO_GRAPH", no_graph)
        pf, qf1, qf2, kw, DDPG, TD3 = _build(kind, 1.0, B)
        pf.load_state_dict(_state(g, f"{kind}_pf0_"))
        qf1.load_state_dict(_state(g, f"{kind}_qf10_"))
        if kind == "ddpg":
            agent = DDPG(pf=pf, qf=qf1, plr=3e-4, qlr=1e-3, **kw)
        else:
            qf2.load_state_dict(_state(g, f"{kind}_qf20_"))
            agent = TD3(pf=pf, qf1=qf1, qf2=qf2, plr=3e-4, qlr=1e-3, policy_update_delay=2, norm_std_policy=0.2,
                        noise_clip=0.5, **kw)
        infos = []
        for s, b in enumerate(batches):
            torch.manual_seed(900 + s)
            infos.append(agent.update(b))
        eng = agent.engine()
        assert len(eng._graphs) == (0 if no_graph == "1" else (1 if kind == "ddpg" else 2))
        assert eng.step_state[:, 0].cpu().tolist() == [float(n) for n in eng.steps]
        results.append((infos, eng.flat.cpu().clone(), eng.tflat.cpu().clone()))
    (ia, fa, ta), (ib, fb, tb) = results
    assert torch.equal(fa, fb) and torch.equal(ta, tb)
    assert ia == ib


def test_off_policy_collector_with_fixed_std_policy_and_short_training():
    """VecCollector + FixGuassianContPolicy (explore = tanh(mlp) + N(0, sigma), CPU draws) fills the replay ring the
    way the oracle collector does; then DDPG / TD3 train for a few updates from random batches of that ring."""
    import torchrl.networks as networks
    import torchrl.policies as policies
    from oracle import nets, replay
    from oracle.detac import det_policy
    from oracle.synth_env import SynthVecEnvCPU
    from torchrl.algo import DDPG, TD3
    from torchrl.collector import VecCollector
    from torchrl.env.synth import SynthVecEnv
    from torchrl.replay_buffers import BaseReplayBuffer
    from torchrl_amd import ops
    N, T, horizon, seed, H, sigma = 32, 12, 5, 2, 64, 0.1
    dev = torch.device(DEV)
    torch.manual_seed(4)
    net = dict(hidden_shapes=[H, H], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.ReLU)
    pf = policies.FixGuassianContPolicy(input_shape=17, output_shape=6, tanh_action=True, norm_std_explore=sigma, **net)
    pf_p = [p.detach().clone() for wb in ops.linear_layers(pf) for p in wb]
    env, eval_env = SynthVecEnv(N, horizon=horizon, device=dev), SynthVecEnv(N, horizon=horizon, device=dev)
    env.seed(seed)
    buf = BaseReplayBuffer(N * T, env_nums=N)
    col = VecCollector(env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=dev, train_render=False,
                       epoch_frames=N * T, max_episode_frames=1000, eval_episodes=1)
    torch.manual_seed(seed)
    state = torch.get_rng_state()
    col.train_one_epoch()

    oenv = SynthVecEnvCPU(N, horizon=horizon)
    oenv.seed(seed)
    ob = oenv.reset()
    torch.set_rng_state(state)
    want_obs, want_act = [], []
    for t in range(T):
        with torch.no_grad():
            a = det_policy(torch.as_tensor(ob, dtype=torch.float32), pf_p, "relu", True) + sigma * torch.randn(N, 6)
        a = a.numpy()
        want_obs.append(np.asarray(ob).copy()); want_act.append(a)
        ob, r, d, info = oenv.step(a)
        if d.any():
            ob = oenv.partial_reset(d.reshape(-1))
    np.testing.assert_allclose(buf._obs.cpu().numpy(), np.stack(want_obs), atol=5e-6)
    np.testing.assert_allclose(buf._acts.cpu().numpy(), np.stack(want_act), atol=5e-6)

    class Stub:
        epoch_frames = N * T
    kw = dict(env=env, replay_buffer=buf, collector=Stub(), logger=_Log(), grad_clip=None, discount=0.99, num_epochs=10,
              batch_size=N * 4, device=dev, save_dir=None, tau=0.005, use_soft_update=True, opt_times=1)
    qf1 = networks.QNet(input_shape=23, output_shape=1, **net)
    qf2 = networks.QNet(input_shape=23, output_shape=1, **net)
    for agent in (DDPG(pf=pf, qf=qf1, plr=3e-4, qlr=1e-3, **kw),
                  TD3(pf=pf, qf1=networks.QNet(input_shape=23, output_shape=1, **net), qf2=qf2, plr=3e-4, qlr=1e-3,
                      noise_mode="device", **kw)):
        before = agent.engine().flat.clone()
        for _ in range(4):
            info = agent.update(buf.random_batch(N * 4, agent.sample_key))
        assert np.isfinite(list(info.values())).all()
        assert (agent.engine().flat - before).abs().max().item() > 0
    ev = col.eval_one_epoch()
    assert len(ev["eval_rewards"]) == N


@pytest.mark.parametrize("algo", ["ddpg", "td3"])
def test_example_scripts_run(tmp_path, algo):
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    params = json.load(open(os.path.join(repo, "config", "%s_synth_halfcheetah.json" % algo)))
    params["net"]["hidden_shapes"] = [64, 64]
    params["replay_buffer"]["size"] = 64 * 64
    params["collector"]["epoch_frames"] = 64 * 8
    params["general_setting"].update(num_epochs=2, batch_size=256, opt_times=3, eval_interval=1, pretrain_epochs=1)
    cfg = tmp_path / ("%s_small.json" % algo)
    cfg.write_text(json.dumps(params))
    out = subprocess.run([sys.executable, os.path.join(repo, "examples", "%s_continuous_vec.py" % algo), "--config", str(cfg),
                          "--vec_env_nums", "64", "--seed", "1", "--log_dir", str(tmp_path / "log"), "--overwrite"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "EPOCH:1" in out.stdout
