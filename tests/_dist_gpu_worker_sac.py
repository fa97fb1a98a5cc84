"""Worker of tests/test_dist_gpu.py (SAC): one rank of an env-sharded twin-Q SAC run, or the single-process run.
argv: rank world port out_path noise_mode."""
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
    if world > 1:
        import torch.distributed as td
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        td.init_process_group("gloo", rank=rank, world_size=world)
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.algo import TwinSACQ
    from torchrl.collector import VecCollector
    from torchrl.env.synth import SynthVecEnv
    from torchrl.replay_buffers import BaseReplayBuffer
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    np.random.seed(0)
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


if __name__ == "__main__":
    main()
