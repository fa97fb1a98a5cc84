"""Host Python envs under the device collectors: `torchrl.env.VecEnv` over single-env
objects with the gym interface, bridged to the same HIP kernels.  The pure-Python stand-in env has the synthetic
dynamics, so the host path must reproduce what the REFERENCE collected for the same seeds
(tests/golden/collect_epoch.npz, obs_norm.npz) and what the on-GPU env path stores."""
import numpy as np
import pytest
import torch

from oracle.synth_env import SynthSingleEnvCPU

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def host_vec_env(N, horizon, seed, alias_reset_obs=False):
    """alias_reset_obs=False: the fixtures of the tests below come from the reference's collectors over a VECTOR env whose
    partial_reset leaves the array `step` returned alone (oracle/synth_env.py), like the on-GPU env -- the reference's own
    VecEnv behaviour (the stored next_obs row of a reset env is the reset observation) is pinned separately, in
    test_reference_vecenv_alias_of_reset_observations below."""
    from torchrl.env import VecEnv
    # the reference's VecEnv.seed gives env i the seed s * N + i (vecenv.py:63-65)
    env = VecEnv(N, [SynthSingleEnvCPU] * N, [(seed * N + i, horizon) for i in range(N)])
    env.alias_reset_obs = alias_reset_obs
    return env


def nets(g, pf_prefix, vf_prefix):
    import torchrl.networks as networks
    import torchrl.policies as policies
    net = dict(hidden_shapes=[64, 64], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.Tanh)
    pf = policies.GuassianContPolicyBasicBias(input_shape=17, output_shape=6, tanh_action=True, **net)
    vf = networks.Net(input_shape=(17,), output_shape=1, **net)
    for prefix, mod in ((pf_prefix, pf), (vf_prefix, vf)):
        mod.load_state_dict({k[len(prefix):].replace("__", "."): torch.tensor(g[k]) for k in g.files if k.startswith(prefix)})
    return pf, vf


@pytest.mark.parametrize("tag", ["small", "surpass", "mixed"])
def test_host_env_collect_matches_reference_and_trains(golden, tag):
    from torchrl.algo import PPO
    from torchrl.collector.on_policy import VecOnPolicyCollector
    from torchrl.replay_buffers.on_policy import OnPolicyReplayBuffer
    from test_product_gpu import ListLogger
    g = golden("collect_epoch")
    N, T, horizon, max_frames, B, seed = (int(x) for x in g[f"{tag}_args"])
    pf, vf = nets(g, tag + "_pf0_", tag + "_vf0_")
    env = host_vec_env(N, horizon, seed)
    buf = OnPolicyReplayBuffer(N * T, env_nums=N, time_limit_filter=True)
    col = VecOnPolicyCollector(vf, env=env, eval_env=host_vec_env(N, horizon, seed + 1), pf=pf, replay_buffer=buf,
                               device=torch.device(DEV), train_render=False, epoch_frames=N * T,
                               max_episode_frames=max_frames, eval_episodes=1)
    assert col.env.is_host_env and col.env.venv is env
    torch.manual_seed(seed)
    res = col.train_one_epoch()
    for k in ("obs", "next_obs", "acts", "values", "rewards", "terminals", "time_limits"):
        err = np.abs(getattr(buf, "_" + k).cpu().numpy() - g[f"{tag}_buf_{k}"]).max()
        assert err < 1e-5, (k, err)
    assert abs(res["train_epoch_reward"] - float(g[f"{tag}_train_epoch_reward"])) < 1e-3
    np.testing.assert_allclose(np.array(res["train_rewards"], dtype=np.float64), g[f"{tag}_train_rewards"], atol=1e-4)
    logger = ListLogger()
    agent = PPO(pf=pf, vf=vf, plr=3e-4, vlr=3e-4, clip_para=0.2, opt_epochs=2, tau=0.95, shuffle=True,
                entropy_coeff=0.005, discount=0.99, num_epochs=10, batch_size=B, gae=True, env=col.env,
                replay_buffer=buf, collector=col, logger=logger, device=torch.device(DEV), save_dir=None)
    p0 = pf.flat_params().clone()
    agent.current_epoch = 0
    agent.update_per_epoch()
    assert logger.infos and all(np.isfinite(list(i.values())).all() for i in logger.infos)
    assert (pf.flat_params() - p0).abs().max() > 0
    ev = col.eval_one_epoch()
    assert len(ev["eval_rewards"]) == N and ev["eval_traj_length"] == horizon


def test_subproc_vecenv_under_the_collector_matches_reference(golden):
    """SubProcVecEnv (2 spawned workers) in place of VecEnv: same collected buffers as the reference."""
    from torchrl.collector.on_policy import VecOnPolicyCollector
    from torchrl.env import SubProcVecEnv
    from torchrl.replay_buffers.on_policy import OnPolicyReplayBuffer
    g = golden("collect_epoch")
    tag = "mixed"
    N, T, horizon, max_frames, B, seed = (int(x) for x in g[f"{tag}_args"])
    pf, vf = nets(g, tag + "_pf0_", tag + "_vf0_")
    procs = 2 if N % 2 == 0 else 1
    env = SubProcVecEnv(procs, N, [SynthSingleEnvCPU] * N, [(seed * N + i, horizon) for i in range(N)])
    env.alias_reset_obs = False                                           # (see host_vec_env)
    try:
        buf = OnPolicyReplayBuffer(N * T, env_nums=N, time_limit_filter=True)
        col = VecOnPolicyCollector(vf, env=env, eval_env=None, pf=pf, replay_buffer=buf, device=torch.device(DEV),
                                   train_render=False, epoch_frames=N * T, max_episode_frames=max_frames, eval_episodes=1)
        torch.manual_seed(seed)
        res = col.train_one_epoch()
        for k in ("obs", "next_obs", "acts", "values", "rewards", "terminals", "time_limits"):
            err = np.abs(getattr(buf, "_" + k).cpu().numpy() - g[f"{tag}_buf_{k}"]).max()
            assert err < 1e-5, (k, err)
        assert abs(res["train_epoch_reward"] - float(g[f"{tag}_train_epoch_reward"])) < 1e-3
        ev = col.eval_one_epoch()                                         # eval env: a deep copy = a second set of workers
        assert len(ev["eval_rewards"]) == N
    finally:
        col.terminate()


@pytest.mark.parametrize("alias", [False, True])
@pytest.mark.parametrize("tag", ["flow", "flow_surpass"])
def test_host_env_with_obs_normaliser_matches_reference(golden, tag, alias):
    """alias: `VecEnv.alias_reset_obs` either way -- behind an observation wrapper the ring row never holds env.step's own
    array (the reference's NormObs.step returns a NEW, normalised array, env/base_wrapper.py:97-121, so its partial_reset
    cannot write into what was stored): the stored rows are the reference's with the flag on (the default) and off."""
    from torchrl.collector.on_policy import VecOnPolicyCollector
    from torchrl.env import HostEnvBridge, NormObs
    from torchrl.replay_buffers.on_policy import OnPolicyReplayBuffer
    g = golden("obs_norm")
    N, T, horizon, max_frames, seed = (int(v) for v in g[tag + "_args"])
    pf, vf = nets(g, tag + "_pf_", tag + "_vf_")
    env = NormObs(HostEnvBridge(host_vec_env(N, horizon, seed, alias_reset_obs=alias), DEV))
    buf = OnPolicyReplayBuffer(N * T, env_nums=N, time_limit_filter=True)
    col = VecOnPolicyCollector(vf, env=env, eval_env=None, pf=pf, replay_buffer=buf, device=torch.device(DEV),
                               train_render=False, epoch_frames=N * T, max_episode_frames=max_frames, eval_episodes=1)
    np.testing.assert_allclose(col.current_ob.cpu().numpy(), g[tag + "_ob0"], rtol=1e-5, atol=2e-6)
    torch.manual_seed(seed)
    res = col.train_one_epoch()
    for k in ("obs", "next_obs", "acts", "values", "rewards", "terminals", "time_limits"):
        err = np.abs(getattr(buf, "_" + k).cpu().numpy() - g[tag + "_buf_" + k]).max()
        assert err < 2e-5, (k, err)
    np.testing.assert_allclose(env._obs_normalizer.state.cpu().numpy(), g[tag + "_state1"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(col.current_ob.cpu().numpy(), g[tag + "_current_ob"], atol=2e-5)
    assert abs(res["train_epoch_reward"] - float(g[tag + "_train_epoch_reward"])) < 1e-3


def test_host_env_off_policy_collector_matches_device_env():
    """VecCollector (SAC-style policy) on a host VecEnv stores the same replay rows as on the on-GPU env."""
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.collector import VecCollector
    from torchrl.env.synth import SynthVecEnv
    from torchrl.replay_buffers import BaseReplayBuffer
    N, steps, horizon, max_frames, seed = 8, 30, 11, 7, 4
    dev = torch.device(DEV)
    net = dict(hidden_shapes=[32, 32], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.ReLU)
    torch.manual_seed(0)
    pf = policies.GuassianContPolicy(input_shape=17, output_shape=12, tanh_action=True, **net)
    rows = {}
    for kind in ("device", "host"):
        if kind == "device":
            env = SynthVecEnv(N, horizon=horizon, device=dev)
            env.seed(seed)
            eval_env = SynthVecEnv(N, horizon=horizon, device=dev)
        else:
            env, eval_env = host_vec_env(N, horizon, seed), host_vec_env(N, horizon, seed + 1)
        buf = BaseReplayBuffer(N * steps, env_nums=N)
        col = VecCollector(env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=dev, train_render=False,
                           epoch_frames=N * steps, max_episode_frames=max_frames, eval_episodes=1)
        torch.manual_seed(seed)
        res = col.train_one_epoch()
        rows[kind] = ({k: getattr(buf, "_" + k).cpu().numpy().copy() for k in
                       ("obs", "next_obs", "acts", "rewards", "terminals", "time_limits")}, res)
        ev = col.eval_one_epoch()
        assert len(ev["eval_rewards"]) == N and ev["eval_traj_length"] == horizon
    for k, v in rows["device"][0].items():
        np.testing.assert_allclose(rows["host"][0][k], v, atol=2e-6, err_msg=k)
    assert abs(rows["host"][1]["train_epoch_reward"] - rows["device"][1]["train_epoch_reward"]) < 1e-3
    np.testing.assert_allclose(rows["host"][1]["train_rewards"], rows["device"][1]["train_rewards"], atol=1e-4)


def test_host_env_example_runs(tmp_path):
    """examples/ppo_host_env.py: VecEnv over a pure-Python pendulum (3 observations, 1 action -> the arbitrary-shape
    PPO engine), a few epochs, snapshots written."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    params = json.load(open(os.path.join(repo, "config", "ppo_pendulum_host.json")))
    params["replay_buffer"]["size"] = params["collector"]["epoch_frames"] = 8 * 208    # 200-step episodes finish
    params["general_setting"].update(num_epochs=3, batch_size=8 * 52, eval_interval=2)
    params["ppo"]["opt_epochs"] = 2
    cfg = tmp_path / "pendulum_small.json"
    cfg.write_text(json.dumps(params))
    out = subprocess.run([sys.executable, os.path.join(repo, "examples", "ppo_host_env.py"), "--config", str(cfg),
                          "--vec_env_nums", "8", "--seed", "1", "--log_dir", str(tmp_path / "log"), "--overwrite"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "EPOCH:2" in out.stdout             # (log_std/std is NaN by construction: unbiased std over ONE action dim)


def test_sac_host_env_example_learns(tmp_path):
    """examples/sac_host_env.py: twin-Q SAC (grouped launches, graph-replayed update) on the pure-Python pendulum through
    VecEnv.  A real task end to end: the greedy return must improve well beyond what a random policy gets (~ -1200)."""
    import json
    import os
    import re
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = os.path.join(repo, "config", "sac_pendulum_host.json")
    assert json.load(open(cfg))["general_setting"]["num_epochs"] == 20
    out = subprocess.run([sys.executable, os.path.join(repo, "examples", "sac_host_env.py"), "--config", cfg,
                          "--vec_env_nums", "8", "--seed", "0", "--log_dir", str(tmp_path / "log"), "--overwrite"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    evals = [float(v) for v in re.findall(r"^Running_Average_Rewards\s+(-?[0-9.]+)", out.stdout, flags=re.M)]
    assert len(evals) >= 5 and evals[-1] > -700.0 and evals[-1] > evals[0] + 400.0, evals


def test_dqn_state_vector_example_learns(tmp_path):
    """examples/dqn_state_vec.py (the reference's script of that name, config/dqn_cartpole.json hyper-parameters): DQN
    with an MLP Q-network on a pure-Python cart-pole through VecEnv (discrete actions).  Random play lasts ~10-20
    steps; the greedy policy must get well past that."""
    import os
    import re
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(repo, "examples", "dqn_state_vec.py"), "--config",
                          os.path.join(repo, "config", "dqn_cartpole_host.json"), "--vec_env_nums", "8", "--seed", "0",
                          "--log_dir", str(tmp_path / "log"), "--overwrite"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    evals = [float(v) for v in re.findall(r"^Running_Average_Rewards\s+(-?[0-9.]+)", out.stdout, flags=re.M)]
    assert len(evals) >= 5 and max(evals[-3:]) > 80.0, evals


def test_ppo_host_env_example_learns(tmp_path):
    """examples/ppo_host_env.py at its shipped config: PPO (arbitrary-shape engine: 3 observations, 1 action) on the
    pure-Python pendulum; the greedy return climbs from the random-policy level (~ -1200) past -600."""
    import os
    import re
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(repo, "examples", "ppo_host_env.py"), "--config",
                          os.path.join(repo, "config", "ppo_pendulum_host.json"), "--vec_env_nums", "16", "--seed", "0",
                          "--log_dir", str(tmp_path / "log"), "--overwrite"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    evals = [float(v) for v in re.findall(r"^Running_Average_Rewards\s+(-?[0-9.]+)", out.stdout, flags=re.M)]
    assert len(evals) >= 5 and evals[-1] > -600.0 and evals[-1] > evals[0] + 400.0, evals


def test_subproc_example_script_runs(tmp_path):
    """examples/ppo_continuous_vec_subproc.py (reference wiring: get_subprocvec_env(..., 4)) on the host pendulum id."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    params = json.load(open(os.path.join(repo, "config", "ppo_pendulum_host.json")))
    params["replay_buffer"]["size"] = params["collector"]["epoch_frames"] = 8 * 208
    params["general_setting"].update(num_epochs=2, batch_size=8 * 52, eval_interval=1)
    params["ppo"]["opt_epochs"] = 2
    cfg = tmp_path / "pendulum_subproc.json"
    cfg.write_text(json.dumps(params))
    out = subprocess.run([sys.executable, os.path.join(repo, "examples", "ppo_continuous_vec_subproc.py"), "--config", str(cfg),
                          "--vec_env_nums", "8", "--seed", "1", "--log_dir", str(tmp_path / "log"), "--overwrite"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "EPOCH:1" in out.stdout


def test_single_env_sac_example_runs(tmp_path):
    """examples/twin_sac_q_continuous.py: `get_env` + `BaseCollector` + `BaseReplayBuffer(size)` (one env)."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    params = json.load(open(os.path.join(repo, "config", "sac_pendulum_single.json")))
    params["general_setting"].update(num_epochs=2, pretrain_epochs=2, opt_times=5, eval_interval=1)
    cfg = tmp_path / "sac_single.json"
    cfg.write_text(json.dumps(params))
    out = subprocess.run([sys.executable, os.path.join(repo, "examples", "twin_sac_q_continuous.py"), "--config", str(cfg),
                          "--seed", "1", "--log_dir", str(tmp_path / "log"), "--overwrite"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "EPOCH:1" in out.stdout


def _reference_style_env(kind, N, horizon, seed):
    from torchrl.env import VecEnv
    from torchrl.env.py_envs import PendulumEnv
    if kind == "synth":                                                       # horizons 4 / 6 / 8 by env index
        return VecEnv(N, [SynthSingleEnvCPU] * N, [(seed * N + i, horizon + 2 * (i % 3)) for i in range(N)])

    class ShortPendulum(PendulumEnv):                                          # (as in tests/golden/make_golden.py)
        def seed(self, seed):
            super().seed(seed)
            self._max_episode_steps = 4 + 2 * (int(seed) % 3)
    env = VecEnv(N, ShortPendulum if kind == "short_pendulum" else PendulumEnv, ())
    env.seed(seed)
    return env


def _state(g, prefix, mod):
    mod.load_state_dict({k[len(prefix):].replace("__", "."): torch.tensor(g[k]) for k in g.files if k.startswith(prefix)})
    return mod


@pytest.mark.parametrize("tag,kind", [("off_pendulum_overlength", "pendulum"), ("off_pendulum_mixed", "short_pendulum"),
                                      ("off_synth_wrap", "synth")])
def test_reference_vecenv_alias_of_reset_observations_off_policy(golden, tag, kind):
    """SURVEY 8(a) a21, VERDICT r04 missing #2: the reference's VecEnv.partial_reset mutates the array `step` returned
    (env/vecenv.py:47-51) and VecCollector.take_actions adds the sample after the reset (collector/base.py:203-227), so
    the ring's `next_obs` rows of reset envs hold the RESET observation -- for over-length resets with terminals False.
    tests/golden/collect_hostenv.npz is the reference's own collectors over the reference's own VecEnv; the default
    (`alias_reset_obs = True`) must store exactly that, and the opt-out exactly the env's own observations."""
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.collector import VecCollector
    from torchrl.replay_buffers import BaseReplayBuffer
    g = golden("collect_hostenv")
    N, steps, rows, horizon, max_frames, seed = (int(v) for v in g[tag + "_args"])
    dev = torch.device(DEV)
    for alias in (True, False):
        env, eval_env = _reference_style_env(kind, N, horizon, seed), _reference_style_env(kind, N, horizon, seed + 1)
        env.alias_reset_obs = alias
        D, A = env.observation_space.shape[0], env.action_space.shape[0]
        net = dict(hidden_shapes=[32, 32], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.ReLU)
        pf = _state(g, tag + "_pf_", policies.GuassianContPolicy(input_shape=D, output_shape=2 * A, tanh_action=True, **net))
        torch.manual_seed(seed)
        buf = BaseReplayBuffer(N * rows, env_nums=N)
        col = VecCollector(env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=dev, train_render=False,
                           epoch_frames=N * steps, max_episode_frames=max_frames, eval_episodes=1)
        np.testing.assert_allclose(col.current_ob.cpu().numpy(), g[tag + "_ob0"], atol=1e-6)
        res = col.train_one_epoch()
        got = {k: getattr(buf, "_" + k).cpu().numpy() for k in ("obs", "next_obs", "acts", "rewards", "terminals", "time_limits")}
        for k in ("obs", "acts", "rewards", "terminals", "time_limits"):
            np.testing.assert_allclose(got[k], np.asarray(g[f"{tag}_buf_{k}"], dtype=np.float64).reshape(got[k].shape),
                                       atol=3e-5, err_msg=k)
        assert [buf._top, buf._size] == list(g[tag + "_top_size"])
        np.testing.assert_allclose(col.current_ob.cpu().numpy(), g[tag + "_current_ob"], atol=3e-5)
        assert abs(res["train_epoch_reward"] - float(g[tag + "_train_epoch_reward"])) < 1e-3
        np.testing.assert_allclose(np.array(res["train_rewards"], dtype=np.float64), g[tag + "_train_rewards"], atol=1e-3)
        want = g[tag + "_buf_next_obs"].astype(np.float64)
        if alias:                                                             # what the reference stores
            np.testing.assert_allclose(got["next_obs"], want, atol=3e-5)
            continue
        # opt-out: the env's own observations -- different from the reference's ring on EXACTLY the reset rows
        true_next, differs = g[tag + "_true_next_obs"], np.zeros((rows, N), dtype=bool)
        for m in g[tag + "_reset_mask"]:
            if int(m[0]) >= steps - rows:
                differs[int(m[0]) % rows] |= m[1:].astype(bool)
        for t in range(max(0, steps - rows), steps):
            np.testing.assert_allclose(got["next_obs"][t % rows], true_next[t], atol=3e-5)
        row_differs = np.abs(got["next_obs"] - want).max(axis=-1) > 1e-4
        written = np.zeros(rows, dtype=bool)
        written[[t % rows for t in range(max(0, steps - rows), steps)]] = True
        assert np.array_equal(row_differs[written], differs[written])
        assert differs.any() and not got["terminals"][differs].all()          # over-length resets: terminals False


@pytest.mark.parametrize("tag,kind", [("on_pendulum_mixed", "short_pendulum"), ("on_synth_mixed", "synth")])
def test_reference_vecenv_alias_of_reset_observations_on_policy(golden, tag, kind):
    """The same for VecOnPolicyCollector (collector/on_policy.py:132-151): the bootstrap value of an over-length env is
    computed from the TRUE next observation (before the reset), the stored row is the reset one."""
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.collector.on_policy import VecOnPolicyCollector
    from torchrl.replay_buffers.on_policy import OnPolicyReplayBuffer
    g = golden("collect_hostenv")
    N, T, horizon, max_frames, seed = (int(v) for v in g[tag + "_args"])
    env, eval_env = _reference_style_env(kind, N, horizon, seed), _reference_style_env(kind, N, horizon, seed + 1)
    D, A = env.observation_space.shape[0], env.action_space.shape[0]
    net = dict(hidden_shapes=[64, 64], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.Tanh)
    pf = _state(g, tag + "_pf_", policies.GuassianContPolicyBasicBias(input_shape=D, output_shape=A, tanh_action=True, **net))
    vf = _state(g, tag + "_vf_", networks.Net(input_shape=(D,), output_shape=1, **net))
    torch.manual_seed(seed)
    buf = OnPolicyReplayBuffer(N * T, env_nums=N, time_limit_filter=True)
    col = VecOnPolicyCollector(vf, env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=torch.device(DEV),
                               train_render=False, epoch_frames=N * T, max_episode_frames=max_frames, eval_episodes=1)
    res = col.train_one_epoch()
    for k in ("obs", "next_obs", "acts", "values", "rewards", "terminals", "time_limits"):
        got = getattr(buf, "_" + k).cpu().numpy()
        np.testing.assert_allclose(got, np.asarray(g[f"{tag}_buf_{k}"], dtype=np.float64).reshape(got.shape), atol=3e-5, err_msg=k)
    np.testing.assert_allclose(col.current_ob.cpu().numpy(), g[tag + "_current_ob"], atol=3e-5)
    assert abs(res["train_epoch_reward"] - float(g[tag + "_train_epoch_reward"])) < 1e-3
    true_next, stored = g[tag + "_true_next_obs"], buf._next_obs.cpu().numpy()
    for m in g[tag + "_reset_mask"]:                                          # the reset rows are NOT the env's own observations
        t, mask = int(m[0]), m[1:].astype(bool)
        assert np.abs(stored[t][mask] - true_next[t][mask]).max() > 1e-4
        np.testing.assert_allclose(stored[t][~mask], true_next[t][~mask], atol=3e-5)


def test_reference_vecenv_alias_with_discrete_actions_and_fallen_poles(golden):
    """The same pin for DISCRETE actions: the reference's VecCollector with an epsilon-greedy Q policy (numpy global stream,
    discrete_policies.py:43-67) over its own VecEnv of cart-poles -- poles that fall (`done`: terminals True) and envs the
    collector resets at max_episode_frames (terminals False, the TD target bootstraps from the stored row) in one ring."""
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.collector import VecCollector
    from torchrl.env import VecEnv
    from torchrl.env.py_envs import CartPoleEnv
    from torchrl.replay_buffers import BaseReplayBuffer
    g = golden("collect_hostenv")
    tag = "off_cartpole_dqn"
    N, steps, rows, max_frames, seed = (int(v) for v in g[tag + "_args"])
    env, eval_env = VecEnv(N, CartPoleEnv, ()), VecEnv(N, CartPoleEnv, ())
    env.seed(seed)
    eval_env.seed(seed + 1)
    qf = _state(g, tag + "_qf_", networks.Net(input_shape=4, output_shape=2, hidden_shapes=[32, 32], append_hidden_shapes=[],
                                               base_type=networks.MLPBase, activation_func=torch.nn.ReLU))
    pf = policies.EpsilonGreedyDQNDiscretePolicy(qf, start_epsilon=0.8, end_epsilon=0.3, decay_frames=25, action_shape=2)
    np.random.seed(seed)
    buf = BaseReplayBuffer(N * rows, env_nums=N)
    col = VecCollector(env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=torch.device(DEV), train_render=False,
                       epoch_frames=N * steps, max_episode_frames=max_frames, eval_episodes=1)
    np.testing.assert_allclose(col.current_ob.cpu().numpy(), g[tag + "_ob0"], atol=1e-6)
    res = col.train_one_epoch()
    for k in ("obs", "next_obs", "acts", "rewards", "terminals", "time_limits"):
        got = getattr(buf, "_" + k).cpu().numpy()[:steps]
        want = np.asarray(g[f"{tag}_buf_{k}"], dtype=np.float64)[:steps].reshape(got.shape)
        np.testing.assert_allclose(got, want, atol=3e-5, err_msg=k)
    assert [buf._top, buf._size] == list(g[tag + "_top_size"])
    assert [pf.epsilon, pf.count] == list(g[tag + "_epsilon_count"])
    np.testing.assert_allclose(col.current_ob.cpu().numpy(), g[tag + "_current_ob"], atol=3e-5)
    np.testing.assert_allclose(np.array(res["train_rewards"], dtype=np.float64), g[tag + "_train_rewards"], atol=1e-6)
    assert abs(res["train_epoch_reward"] - float(g[tag + "_train_epoch_reward"])) < 1e-6
    stored, true_next, term = buf._next_obs.cpu().numpy(), g[tag + "_true_next_obs"], buf._terminals.cpu().numpy().reshape(rows, N)
    over_length = 0
    for m in g[tag + "_reset_mask"]:
        t, mask = int(m[0]), m[1:].astype(bool)
        assert np.abs(stored[t][mask] - true_next[t][mask]).max() > 1e-4        # the reset observation, not the env's own
        np.testing.assert_allclose(stored[t][~mask], true_next[t][~mask], atol=3e-5)
        over_length += int((mask & (term[t] == 0)).sum())
    assert over_length > 0 and term[:steps].sum() > 0


@pytest.mark.parametrize("tag", ["off_subproc_done", "off_subproc_overlength"])
def test_reference_subproc_vecenv_alias_of_reset_observations(golden, tag):
    """The reference's SubProcVecEnv writes the fresh observations into the stacked array `step` returned as well
    (env/subproc_vecenv.py:108-121); under its VecCollector the ring's `next_obs` rows of reset envs are the reset
    observations -- after env `done` and after the collector's own limit.  The product's SubProcVecEnv (spawned workers)
    under the device collector must store the same ring."""
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.collector import VecCollector
    from torchrl.env import SubProcVecEnv
    from torchrl.replay_buffers import BaseReplayBuffer
    g = golden("collect_hostenv")
    N, procs, steps, rows, horizon, max_frames, seed = (int(v) for v in g[tag + "_args"])
    env = SubProcVecEnv(procs, N, [SynthSingleEnvCPU] * N, [(0, horizon)] * N)    # (the reference's workers ignore `seed`: all 0)
    assert env.alias_reset_obs
    col = None
    try:
        net = dict(hidden_shapes=[32, 32], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.ReLU)
        pf = _state(g, tag + "_pf_", policies.GuassianContPolicy(input_shape=17, output_shape=12, tanh_action=True, **net))
        torch.manual_seed(seed)
        buf = BaseReplayBuffer(N * rows, env_nums=N)
        col = VecCollector(env=env, eval_env=None, pf=pf, replay_buffer=buf, device=torch.device(DEV), train_render=False,
                           epoch_frames=N * steps, max_episode_frames=max_frames, eval_episodes=1)
        np.testing.assert_allclose(col.current_ob.cpu().numpy(), g[tag + "_ob0"], atol=1e-6)
        res = col.train_one_epoch()
        for k in ("obs", "next_obs", "acts", "rewards", "terminals", "time_limits"):
            got = getattr(buf, "_" + k).cpu().numpy()[:steps]
            np.testing.assert_allclose(got, np.asarray(g[f"{tag}_buf_{k}"], dtype=np.float64)[:steps].reshape(got.shape),
                                       atol=3e-5, err_msg=k)
        assert [buf._top, buf._size] == list(g[tag + "_top_size"])
        np.testing.assert_allclose(col.current_ob.cpu().numpy(), g[tag + "_current_ob"], atol=3e-5)
        assert abs(res["train_epoch_reward"] - float(g[tag + "_train_epoch_reward"])) < 1e-3
        np.testing.assert_allclose(np.array(res["train_rewards"], dtype=np.float64), g[tag + "_train_rewards"], atol=1e-4)
        stored, true_next = buf._next_obs.cpu().numpy(), g[tag + "_true_next_obs"]
        assert len(g[tag + "_reset_mask"]) >= 2
        for m in g[tag + "_reset_mask"]:
            t, mask = int(m[0]), m[1:].astype(bool)
            assert np.abs(stored[t][mask] - true_next[t][mask]).max() > 1e-4
    finally:
        if col is not None:
            col.terminate()
        else:
            env.close()
