"""A2C on the fused engine: A2C.update against what the REFERENCE's A2C.update
produced (tests/golden/a2c_update.npz), and the epoch loop against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _StubCollector:
    epoch_frames = 0


class ListLogger:
    def __init__(self): self.infos = []
    def add_update_info(self, d): self.infos.append(dict(d))
    def add_epoch_info(self, *a, **k): pass
    def log(self, *a): pass
    def finish(self): pass


def _nets(g, tag):
    import torchrl.networks as networks
    import torchrl.policies as policies
    net = dict(hidden_shapes=[64, 64], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.Tanh)
    pf = policies.GuassianContPolicyBasicBias(input_shape=17, output_shape=6, tanh_action=True, **net)
    vf = networks.Net(input_shape=(17,), output_shape=1, **net)
    for prefix, mod in ((f"{tag}_pf0_", pf), (f"{tag}_vf0_", vf)):
        mod.load_state_dict({k[len(prefix):].replace("__", "."): torch.tensor(g[k]) for k in g.files if k.startswith(prefix)})
    return pf, vf


@pytest.mark.parametrize("engine", ["fused", "generic"])
@pytest.mark.parametrize("tag", ["small", "mid"])
def test_a2c_update_matches_reference(golden, tag, engine, monkeypatch):
    monkeypatch.setenv("TRL_GENERIC_PPO", "1" if engine == "generic" else "0")   # generic: arbitrary-shape engine
    from torchrl.algo import A2C
    from torchrl.env.synth import SynthVecEnv
    g = golden("a2c_update")
    pf, vf = _nets(g, tag)
    dev = torch.device(DEV)
    agent = A2C(pf=pf, vf=vf, plr=3e-4, vlr=1e-3, entropy_coeff=0.01, tau=0.95, shuffle=True, discount=0.99,
                num_epochs=10, batch_size=64, gae=True, env=SynthVecEnv(4, device=dev), replay_buffer=None,
                collector=_StubCollector(), logger=ListLogger(), device=dev, save_dir=None)
    batch = {k: g[f"{tag}_batch_{k}"] for k in ("obs", "acts", "advs", "estimate_returns")}
    for s in range(2):
        info = agent.update(batch)
        keys = [str(k) for k in g[f"{tag}_info{s}_keys"]]
        assert sorted(info.keys()) == keys
        # scalar statistics: rel 1e-4 / abs 1e-5
        np.testing.assert_allclose([info[k] for k in keys], g[f"{tag}_info{s}_vals"], rtol=2e-4, atol=2e-5)
        for prefix, mod in ((f"{tag}_pf{s + 1}_", pf), (f"{tag}_vf{s + 1}_", vf)):
            for name, p in mod.state_dict().items():
                err = np.abs(p.cpu().numpy() - g[prefix + name.replace(".", "__")]).max()
                assert err < 2e-6, (s, name, err)                           # post-step params, abs 1e-6 class
    assert agent.training_update_num == 2


def test_a2c_epoch_matches_oracle():
    """collector -> GAE -> one pass of minibatches (on_rl_algo.py:35-40) on the device vs the CPU oracle."""
    import torchrl.networks as networks
    import torchrl.policies as policies
    from oracle import replay
    from oracle.collector import VecOnPolicyCollectorOracle
    from oracle.ppo import A2COracle
    from oracle.synth_env import SynthVecEnvCPU
    from torchrl.algo import A2C
    from torchrl.collector.on_policy import VecOnPolicyCollector
    from torchrl.env.synth import SynthVecEnv
    from torchrl.replay_buffers.on_policy import OnPolicyReplayBuffer
    N, T, B, seed, horizon = 16, 24, 96, 5, 9
    dev = torch.device(DEV)
    torch.manual_seed(3)
    net = dict(hidden_shapes=[64, 64], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.Tanh)
    pf = policies.GuassianContPolicyBasicBias(input_shape=17, output_shape=6, tanh_action=True, **net)
    vf = networks.Net(input_shape=(17,), output_shape=1, **net)
    pf_p = [p.detach().clone() for p in pf._mlp2_param_list()]
    ls_p = pf.logstd.detach().clone()
    vf_p = [p.detach().clone() for p in vf._mlp2_param_list()]

    env = SynthVecEnv(N, horizon=horizon, device=dev)
    env.seed(seed)
    buf = OnPolicyReplayBuffer(N * T, env_nums=N, time_limit_filter=True)
    col = VecOnPolicyCollector(vf, env=env, eval_env=SynthVecEnv(N, horizon=horizon, device=dev), pf=pf, replay_buffer=buf,
                               device=dev, train_render=False, epoch_frames=N * T, max_episode_frames=1000, eval_episodes=1)
    logger = ListLogger()
    agent = A2C(pf=pf, vf=vf, plr=3e-4, vlr=3e-4, entropy_coeff=0.005, tau=0.95, shuffle=True, discount=0.99,
                num_epochs=10, batch_size=B, gae=True, env=env, replay_buffer=buf, collector=col, logger=logger,
                device=dev, save_dir=None)
    torch.manual_seed(seed)
    noise_state = torch.get_rng_state()
    col.train_one_epoch()
    np.random.seed(seed + 1)
    agent.update_per_epoch()

    cenv = SynthVecEnvCPU(N, horizon=horizon)
    cenv.seed(seed)
    ring = replay.RingOracle(N * T, env_nums=N, time_limit_filter=True)
    ocol = VecOnPolicyCollectorOracle(cenv, ring, pf_p, ls_p, vf_p, epoch_frames=N * T, max_episode_frames=1000)
    torch.set_rng_state(noise_state)
    ocol.train_one_epoch()
    o = A2COracle(pf_p, ls_p, vf_p, plr=3e-4, vlr=3e-4, entropy_coeff=0.005, batch_size=B, discount=0.99, tau=0.95)
    np.random.seed(seed + 1)
    want = o.epoch(ring)
    assert len(logger.infos) == len(want) == T * N // B
    keys = sorted(want[0].keys())
    got = np.array([[i[k] for k in keys] for i in logger.infos])
    np.testing.assert_allclose(got, np.array([[i[k] for k in keys] for i in want]), rtol=3e-4, atol=5e-5)
    for a, b in zip(pf._mlp2_param_list() + [pf.logstd] + vf._mlp2_param_list(), o.pf + [o.logstd] + o.vf):
        assert (a.detach().cpu() - b.detach()).abs().max().item() < 3e-6


def test_a2c_example_script_runs(tmp_path):
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    params = json.load(open(os.path.join(repo, "config", "a2c_synth_halfcheetah.json")))
    params["replay_buffer"]["size"] = 64 * 32
    params["collector"]["epoch_frames"] = 64 * 32
    params["general_setting"].update(num_epochs=2, batch_size=512, eval_interval=1)
    cfg = tmp_path / "a2c_small.json"
    cfg.write_text(json.dumps(params))
    out = subprocess.run([sys.executable, os.path.join(repo, "examples", "a2c_continuous_vec.py"), "--config", str(cfg),
                          "--vec_env_nums", "64", "--seed", "1", "--log_dir", str(tmp_path / "log"), "--overwrite"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "EPOCH:1" in out.stdout
