"""Greedy evaluation loop on the device against what the REFERENCE's
VecCollector.eval_one_epoch returned (torchrl/collector/base.py:232-280; tests/golden/eval_epoch.npz)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def state(g, prefix):
    return {k[len(prefix):].replace("__", "."): torch.tensor(g[k]) for k in g.files if k.startswith(prefix)}


def test_eval_loop_on_the_synthetic_env_matches_reference(golden):
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.collector import VecCollector
    from torchrl.env.synth import SynthVecEnv
    from torchrl.replay_buffers import BaseReplayBuffer
    g = golden("eval_epoch")
    N, horizon, episodes, seed = (int(x) for x in g["synth_args"])
    dev = torch.device(DEV)
    net = dict(hidden_shapes=[32, 32], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.ReLU)
    pf = policies.GuassianContPolicy(input_shape=17, output_shape=12, tanh_action=True, **net)
    pf.load_state_dict(state(g, "synth_pf_"))
    env, eval_env = SynthVecEnv(N, horizon=horizon, device=dev), SynthVecEnv(N, horizon=horizon, device=dev)
    env.seed(seed)
    eval_env.seed(seed + 1)
    col = VecCollector(env=env, eval_env=eval_env, pf=pf, replay_buffer=BaseReplayBuffer(N * 4, env_nums=N), device=dev,
                       epoch_frames=N * 4, max_episode_frames=999, eval_episodes=episodes)
    res = col.eval_one_epoch()
    np.testing.assert_allclose(np.array(res["eval_rewards"], dtype=np.float64).reshape(-1), g["synth_eval_rewards"], atol=5e-5)
    assert res["eval_traj_length"] == float(g["synth_eval_traj_length"])


def test_eval_loop_over_host_envs_matches_reference(golden):
    """Episodes of different lengths: every env's FIRST episode counts, finished envs are reset and keep stepping until
    the last one is done -- the cart-pole's reset draws make the second round depend on exactly that."""
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.collector import VecCollector
    from torchrl.env import VecEnv
    from torchrl.env.py_envs import CartPoleEnv
    from torchrl.replay_buffers import BaseReplayBuffer
    g = golden("eval_epoch")
    N, H, episodes, seed = (int(x) for x in g["cartpole_args"])
    dev = torch.device(DEV)
    env, eval_env = VecEnv(N, CartPoleEnv, ()), VecEnv(N, CartPoleEnv, ())
    env.seed(seed)
    eval_env.seed(seed + 1)
    qf = networks.Net(input_shape=4, output_shape=2, hidden_shapes=[H, H], append_hidden_shapes=[],
                      base_type=networks.MLPBase, activation_func=torch.nn.ReLU)
    qf.load_state_dict(state(g, "cartpole_qf_"))
    pf = policies.EpsilonGreedyDQNDiscretePolicy(qf, start_epsilon=1.0, end_epsilon=0.05, decay_frames=1000, action_shape=2)
    col = VecCollector(env=env, eval_env=eval_env, pf=pf, replay_buffer=BaseReplayBuffer(N * 4, env_nums=N), device=dev,
                       epoch_frames=N * 4, max_episode_frames=999, eval_episodes=episodes)
    res = col.eval_one_epoch()
    np.testing.assert_array_equal(np.array(res["eval_rewards"], dtype=np.float64).reshape(-1), g["cartpole_eval_rewards"])
    assert abs(res["eval_traj_length"] - float(g["cartpole_eval_traj_length"])) < 1e-9


def test_eps_greedy_explore_matches_reference(golden):
    """EpsilonGreedyDQNDiscretePolicy.explore (discrete_policies.py:43-67): epsilon schedule, host numpy draws and the
    argmax + mask kernel reproduce the reference's exploration decisions call by call."""
    import torchrl.networks as networks
    import torchrl.policies as policies
    g = golden("eps_greedy")
    N, A, D, H, calls, decay = (int(x) for x in g["args"])
    qf = networks.Net(input_shape=D, output_shape=A, hidden_shapes=[H, H], append_hidden_shapes=[],
                      base_type=networks.MLPBase, activation_func=torch.nn.ReLU)
    qf.load_state_dict(state(g, "qf_"))
    qf.to(DEV)
    pf = policies.EpsilonGreedyDQNDiscretePolicy(qf, start_epsilon=0.9, end_epsilon=0.15, decay_frames=decay, action_shape=A)
    np.random.seed(21)
    for c in range(calls):
        out = pf.explore(torch.tensor(g["obs"][c], device=DEV).unsqueeze(0))
        assert out["action"].shape == (N, 1) and out["action"].dtype == torch.int64
        assert np.array_equal(out["action"].cpu().numpy(), g["actions"][c]), c
        assert pf.epsilon == float(g["epsilon"][c])

