"""Leaving the HIP path must be loud: whole PPO / SAC / DQN epochs (collect -> sample -> update, plus evaluation) run
without a single CUDA tensor taking the nn.Module / torch.distributions route (torchrl_amd._C.note_eager counts
them), and the counter does trip -- and TRL_STRICT=1 raises -- when a caller forces that route."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _epoch(agent, col):
    res = col.train_one_epoch()
    agent.current_epoch = 0
    agent.update_per_epoch()
    col.eval_one_epoch()
    torch.cuda.synchronize()
    return res


def test_ppo_epoch_fused_and_generic_shapes_stay_on_the_hip_path(monkeypatch):
    from torchrl_amd import _C
    from tests.test_fullsize_gpu import make
    before = _C.eager_fallback_count()
    pf, vf, env, buf, col, agent = make(64)
    np.random.seed(0)
    _epoch(agent, col)
    monkeypatch.setenv("TRL_GENERIC_PPO", "1")                       # the arbitrary-shape engine on the same networks
    pf, vf, env, buf, col, agent = make(64)
    np.random.seed(0)
    _epoch(agent, col)
    assert type(agent.engine()).__name__ == "_GenericPPO"
    assert _C.eager_fallback_count() == before, _C.EAGER_FALLBACKS


def test_sac_epoch_stays_on_the_hip_path():
    from torchrl_amd import _C
    from tests.test_fullsize_offpolicy_gpu import build_cfg3
    before = _C.eager_fallback_count()
    pf, qf1, qf2, env, buf, col, agent, log = build_cfg3(64, 0, 64)
    np.random.seed(0)
    torch.manual_seed(0)
    for _ in range(3):                                               # eager, captured, replayed
        _epoch(agent, col)
    assert len(log.infos) == 3
    assert _C.eager_fallback_count() == before, _C.EAGER_FALLBACKS


@pytest.mark.parametrize("Q", [1, 8])
def test_dqn_epoch_stays_on_the_hip_path(Q):
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.algo import DQN, QRDQN
    from torchrl.collector import VecCollector
    from torchrl.env import get_vec_env
    from torchrl.replay_buffers import BaseReplayBuffer
    from torchrl_amd import _C
    from tests.test_fullsize_offpolicy_gpu import CONVS, _Log
    before = _C.eager_fallback_count()
    N, A = 16, 6
    torch.manual_seed(0)
    qf = networks.Net(output_shape=A * Q, base_type=networks.CNNBase, append_hidden_shapes=[512],
                      activation_func=torch.nn.Tanh, input_shape=(4, 84, 84), hidden_shapes=CONVS)
    env = get_vec_env("SynthAtari-v0", {"reward_scale": 1}, N)
    eval_env = get_vec_env("SynthAtari-v0", {"reward_scale": 1}, N)
    env.horizon = eval_env.horizon = 3
    kwp = dict(qf=qf, start_epsilon=1, end_epsilon=0.1, decay_frames=1000, action_shape=A)
    pf = policies.EpsilonGreedyQRDQNDiscretePolicy(quantile_num=Q, **kwp) if Q > 1 else policies.EpsilonGreedyDQNDiscretePolicy(**kwp)
    buf = BaseReplayBuffer(8 * N, env_nums=N)
    col = VecCollector(env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=DEV, epoch_frames=N * 4,
                       max_episode_frames=999)
    kw = dict(qf=qf, pf=pf, qlr=2.5e-4, env=env, replay_buffer=buf, collector=col, logger=_Log(), discount=0.99,
              num_epochs=1, batch_size=2 * N, device=DEV, save_dir=None, tau=0.005, opt_times=2)
    agent = QRDQN(quantile_num=Q, **kw) if Q > 1 else DQN(**kw)
    np.random.seed(0)
    _epoch(agent, col)
    assert _C.eager_fallback_count() == before, _C.EAGER_FALLBACKS


def test_forced_eager_route_is_counted_and_strict_mode_raises(monkeypatch):
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl_amd import _C
    net = dict(hidden_shapes=[64, 64], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.Tanh)
    pf = policies.GuassianContPolicyBasicBias(input_shape=17, output_shape=6, tanh_action=True, **net).to(DEV)
    x = torch.randn(4, 17, device=DEV)
    before = _C.eager_fallback_count()
    with torch.no_grad():
        pf(x)                                                        # HIP kernel: not counted
    assert _C.eager_fallback_count() == before
    pf(x)                                                            # autograd on: nn.Module graph, counted
    assert _C.eager_fallback_count() == before + 1
    out = pf.update(x, torch.tanh(torch.randn(4, 6, device=DEV)))    # torch.distributions log-prob with autograd
    assert out["log_prob"].requires_grad and _C.eager_fallback_count() >= before + 3
    monkeypatch.setenv("TRL_STRICT", "1")
    with pytest.raises(_C.TrlError, match="left the HIP path"):
        pf(x)
    with torch.no_grad():
        pf(x)                                                        # the HIP route is unaffected
