"""Worker of tests/test_dist_gpu.py (off-policy algorithms): one rank of an env-sharded twin-Q SAC / TD3 / DQN run, or the
single-process run.  argv: rank world port out_path noise_mode [sac|td3|dqn]."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

N_TOTAL, STEPS, ROWS_B, EPOCHS, OPT = 32, 12, 4, 3, 4


class Log:
    def __init__(self): self.infos = []
    def add_update_info(self, d): self.infos.append(dict(d))
    def add_epoch_info(self, *a, **k): pass
    def log(self, *a): pass
    def finish(self): pass


def main():
    rank, world, port, out, noise = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    algo_name = sys.argv[6] if len(sys.argv) > 6 else "sac"
    if world > 1:
        import torch.distributed as td
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        td.init_process_group("gloo", rank=rank, world_size=world)
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.algo import DQN, TD3, TwinSACQ
    from torchrl.collector import VecCollector
    from torchrl.env.synth import SynthFrameVecEnv, SynthVecEnv
    from torchrl.replay_buffers import BaseReplayBuffer
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    np.random.seed(0)
    if algo_name != "sac":
        return other(algo_name, rank, world, out, noise, dev)
    net = dict(hidden_shapes=[64, 64], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.ReLU)
    pf = policies.GuassianContPolicy(input_shape=17, output_shape=12, tanh_action=True, **net)
    qf1 = networks.QNet(input_shape=23, output_shape=1, **net)
    qf2 = networks.QNet(input_shape=23, output_shape=1, **net)
    n = N_TOTAL // world
    kw = dict(horizon=10, device=dev, index_offset=rank * n, total_env_nums=N_TOTAL)
    env, eval_env = SynthVecEnv(n, **kw), SynthVecEnv(n, **kw)
    env.seed(2)
    buf = BaseReplayBuffer(n * 64, env_nums=n)
    col = VecCollector(env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=dev, train_render=False,
                       epoch_frames=n * STEPS, max_episode_frames=8, eval_episodes=1, noise_mode=noise)
    logger = Log()
    agent = TwinSACQ(pf=pf, qf1=qf1, qf2=qf2, plr=3e-4, qlr=1e-3, policy_std_reg_weight=1e-3, policy_mean_reg_weight=1e-3,
                     reparameterization=True, automatic_entropy_tuning=True, noise_mode=noise, env=env, replay_buffer=buf,
                     collector=col, logger=logger, grad_clip=1.0, discount=0.99, num_epochs=10, batch_size=ROWS_B * n,
                     device=dev, save_dir=None, tau=0.005, use_soft_update=True, opt_times=OPT)
    torch.manual_seed(7)
    for epoch in range(EPOCHS):
        col.train_one_epoch()
        agent.update_per_epoch()
    keys = sorted(logger.infos[0])
    eng = agent.engine()
    np.savez(out, flat=eng.flat.cpu().numpy(), tflat=eng.tflat.cpu().numpy(), log_alpha=agent.log_alpha.cpu().numpy(),
             keys=np.array(keys), infos=np.array([[i[k] for k in keys] for i in logger.infos]),
             obs=buf._obs.cpu().numpy(), acts=buf._acts.cpu().numpy())
    if world > 1:
        td.barrier()
        td.destroy_process_group()


def other(algo_name, rank, world, out, noise, dev):
    import torch.distributed as td
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.algo import DQN, TD3
    from torchrl.collector import VecCollector
    from torchrl.env.synth import SynthFrameVecEnv, SynthVecEnv
    from torchrl.replay_buffers import BaseReplayBuffer
    n = N_TOTAL // world
    logger = Log()
    common = dict(logger=logger, discount=0.99, num_epochs=10, batch_size=ROWS_B * n, device=dev, save_dir=None, tau=0.005,
                  use_soft_update=True, opt_times=OPT)
    if algo_name == "td3":
        net = dict(hidden_shapes=[64, 64], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.ReLU)
        pf = policies.FixGuassianContPolicy(input_shape=17, output_shape=6, tanh_action=True, norm_std_explore=0.1, **net)
        qf1 = networks.QNet(input_shape=23, output_shape=1, **net)
        qf2 = networks.QNet(input_shape=23, output_shape=1, **net)
        kw = dict(horizon=10, device=dev, index_offset=rank * n, total_env_nums=N_TOTAL)
        env, eval_env = SynthVecEnv(n, **kw), SynthVecEnv(n, **kw)
        env.seed(2)
        buf = BaseReplayBuffer(n * 64, env_nums=n)
        col = VecCollector(env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=dev, train_render=False,
                           epoch_frames=n * STEPS, max_episode_frames=8, eval_episodes=1, noise_mode=noise)
        agent = TD3(pf=pf, qf1=qf1, qf2=qf2, plr=3e-4, qlr=1e-3, policy_update_delay=2, norm_std_policy=0.2, noise_clip=0.5,
                    env=env, replay_buffer=buf, collector=col, grad_clip=1.0, **common)
        agent.noise_mode = noise
    else:
        convs = [[8, [8, 8], [4, 4], [0, 0]], [8, [4, 4], [2, 2], [0, 0]], [16, [3, 3], [1, 1], [0, 0]]]
        qf = networks.Net(output_shape=6, base_type=networks.CNNBase, append_hidden_shapes=[32], activation_func=torch.nn.Tanh,
                          input_shape=(4, 84, 84), hidden_shapes=convs)
        pf = policies.EpsilonGreedyDQNDiscretePolicy(qf=qf, start_epsilon=0.5, end_epsilon=0.1, decay_frames=30, action_shape=6)
        kw = dict(horizon=10, device=dev, index_offset=rank * n, total_env_nums=N_TOTAL)
        env, eval_env = SynthFrameVecEnv(n, **kw), SynthFrameVecEnv(n, **kw)
        env.seed(2)
        buf = BaseReplayBuffer(n * 32, env_nums=n)
        col = VecCollector(env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=dev, train_render=False,
                           epoch_frames=n * STEPS, max_episode_frames=8, eval_episodes=1)
        agent = DQN(pf=pf, qf=qf, qlr=1e-3, env=env, replay_buffer=buf, collector=col, **common)
    torch.manual_seed(7)
    for epoch in range(EPOCHS):
        col.train_one_epoch()
        agent.update_per_epoch()
    keys = sorted(k for k in logger.infos[-1] if all(k in i for i in logger.infos))
    eng = agent.engine()
    np.savez(out, flat=eng.flat.cpu().numpy(), tflat=eng.tflat.cpu().numpy(), keys=np.array(keys),
             infos=np.array([[i[k] for k in keys] for i in logger.infos]), acts=buf._acts.cpu().numpy(),
             rewards=buf._rewards.cpu().numpy())
    if world > 1:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
