"""V-MPO on the HIP path against what the REFERENCE's VMPO.update produced for the same batches and parameters
(tests/golden/vmpo_update.npz): info dicts, eta / alpha, post-step parameters; then an end-to-end epoch."""
import numpy as np
import pytest
import torch

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


def _nets(g, tag, D, A, H):
    import torchrl.networks as networks
    import torchrl.policies as policies
    net = dict(hidden_shapes=[H, H], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.Tanh)
    pf = policies.GuassianContPolicyBasicBias(input_shape=D, output_shape=A, tanh_action=True, **net)
    vf = networks.Net(input_shape=(D,), output_shape=1, **net)
    for prefix, mod in ((f"{tag}_pf0_", pf), (f"{tag}_vf0_", vf)):
        mod.load_state_dict({k[len(prefix):].replace("__", "."): torch.tensor(g[k]) for k in g.files if k.startswith(prefix)})
    return pf, vf


@pytest.mark.parametrize("tag", ["small", "odd"])
def test_vmpo_update_matches_reference(golden, tag):
    from torchrl.algo import VMPO
    from torchrl.env.synth import SynthVecEnv
    g = golden("vmpo_update")
    B, D, A, H = (int(v) for v in g[tag + "_args"])
    pf, vf = _nets(g, tag, D, A, H)
    agent = VMPO(pf=pf, vf=vf, plr=1e-3, vlr=1e-3, opt_epochs=2, eta_eps=0.02, alpha_eps=0.1, tau=0.95, shuffle=True,
                 discount=0.99, num_epochs=10, batch_size=B, gae=True, env=SynthVecEnv(4, obs_dim=D, act_dim=A, device=DEV),
                 replay_buffer=None, collector=_Stub(), logger=_Log(), device=DEV, save_dir=None)
    agent.engine().sync_target_pf()
    for s in range(3):
        batch = {k: g[f"{tag}_s{s}_batch_{k}"] for k in ("obs", "acts", "advs", "values", "estimate_returns")}
        info = agent.update(batch)
        keys = [str(k) for k in g[f"{tag}_s{s}_info_keys"]]
        assert sorted(info.keys()) == keys
        # scalar statistics: rel 2e-4 / abs 2e-5; KL values are O(1e-3) differences of O(1) terms
        np.testing.assert_allclose([info[k] for k in keys], g[f"{tag}_s{s}_info_vals"], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose([agent.eta.item(), agent.alpha.item()], g[f"{tag}_s{s}_eta_alpha"], rtol=2e-6)
    for prefix, mod in ((f"{tag}_pf1_", pf), (f"{tag}_vf1_", vf)):
        for name, p in mod.state_dict().items():
            err = np.abs(p.cpu().numpy() - g[prefix + name.replace(".", "__")]).max()
            assert err < 3e-6, (name, err)
    assert agent.training_update_num == 3


def test_vmpo_epoch_runs_on_the_collector():
    """collector -> GAE -> opt_epochs passes of one_iteration minibatches (v_mpo.py:44-56), device env."""
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.algo import VMPO
    from torchrl.collector.on_policy import VecOnPolicyCollector
    from torchrl.env.synth import SynthVecEnv
    from torchrl.replay_buffers.on_policy import OnPolicyReplayBuffer
    N, T = 32, 16
    net = dict(hidden_shapes=[64, 64], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.Tanh)
    pf = policies.GuassianContPolicyBasicBias(input_shape=17, output_shape=6, tanh_action=True, **net)
    vf = networks.Net(input_shape=(17,), output_shape=1, **net)
    env, eval_env = SynthVecEnv(N, horizon=12, device=DEV), SynthVecEnv(N, horizon=12, device=DEV)
    env.seed(5)
    buf = OnPolicyReplayBuffer(N * T, env_nums=N, time_limit_filter=True)
    col = VecOnPolicyCollector(vf, env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=DEV, train_render=False,
                               epoch_frames=N * T, max_episode_frames=9, eval_episodes=1, noise_mode="device")
    logger = _Log()
    agent = VMPO(pf=pf, vf=vf, plr=3e-4, vlr=3e-4, opt_epochs=2, tau=0.95, shuffle=True, discount=0.99, num_epochs=10,
                 batch_size=N * 4, gae=True, env=env, replay_buffer=buf, collector=col, logger=logger, device=DEV,
                 save_dir=None)
    p0 = torch.cat([p.detach().reshape(-1) for p in pf.parameters()]).clone()
    for epoch in range(2):
        col.train_one_epoch()
        agent.current_epoch = epoch
        agent.update_per_epoch()
    assert len(logger.infos) == 2 * 2 * (T // 4)
    assert all(np.isfinite(list(i.values())).all() for i in logger.infos)
    assert logger.infos[0]["KL/max"] == 0.0                             # first minibatch of an epoch: pi == pi_target
    assert (torch.cat([p.detach().reshape(-1) for p in pf.parameters()]) - p0).abs().max() > 0
    assert 0 < agent.alpha.item() < 0.2 and 0.5 < agent.eta.item() < 1.5



def test_vmpo_example_script_runs(tmp_path):
    """examples/vmpo_continuous_vec.py (reference wiring, its config/vmpo_halfcheetah.json hyper-parameters incl. obs_norm)."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    params = json.load(open(os.path.join(repo, "config", "vmpo_synth_halfcheetah.json")))
    params["general_setting"].update(num_epochs=3, eval_interval=1)
    params["vmpo"]["opt_epochs"] = 2
    cfg = tmp_path / "vmpo_small.json"
    cfg.write_text(json.dumps(params))
    out = subprocess.run([sys.executable, os.path.join(repo, "examples", "vmpo_continuous_vec.py"), "--config", str(cfg),
                          "--vec_env_nums", "16", "--seed", "1", "--log_dir", str(tmp_path / "log"), "--overwrite"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "EPOCH:2" in out.stdout and "Training/eta" in out.stdout
