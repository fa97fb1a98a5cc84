"""Twin-Q SAC on ONE env -- the wiring of the reference's examples/twin_sac_q_continuous.py (`get_env`, `BaseCollector`, a
replay buffer without `env_nums`), here on the host pendulum id so that it learns something:

    python examples/twin_sac_q_continuous.py --config config/sac_pendulum_single.json --seed 0 --overwrite
"""
import os.path as osp
import random
import sys

import numpy as np
import torch

sys.path.append(osp.join(osp.dirname(osp.abspath(__file__)), ".."))
import torchrl.networks as networks                       # noqa: E402
import torchrl.policies as policies                       # noqa: E402
from torchrl.algo import TwinSACQ                         # noqa: E402
from torchrl.collector.base import BaseCollector              # noqa: E402
from torchrl.env import get_env                           # noqa: E402
from torchrl.replay_buffers import BaseReplayBuffer       # noqa: E402
from torchrl.utils import Logger, get_args, get_params    # noqa: E402


def main():
    args = get_args()
    params = get_params(args.config)
    device = torch.device("cuda:{}".format(args.device) if args.cuda else "cpu")
    if device.type == "cuda":
        torch.cuda.set_device(device)                    # envs / replay buffers allocate on the current device
    env = get_env(params['env_name'], params['env'])
    env.seed(args.seed)
    for seed_fn in (torch.manual_seed, np.random.seed, random.seed):
        seed_fn(args.seed)
    name = args.id if args.id is not None else osp.splitext(osp.basename(args.config))[0]
    logger = Logger(name, params['env_name'], args.seed, params, args.log_dir, args.overwrite)
    replay_buffer = BaseReplayBuffer(max_replay_buffer_size=int(params['replay_buffer']['size']),
                                     time_limit_filter=params['replay_buffer']['time_limit_filter'])
    net = dict(params['net'], base_type=networks.MLPBase, activation_func=torch.nn.ReLU)
    obs_dim, act_dim = env.observation_space.shape[0], env.action_space.shape[0]
    pf = policies.GuassianContPolicy(input_shape=obs_dim, output_shape=2 * act_dim, **net, **params['policy'])
    qf1 = networks.QNet(input_shape=obs_dim + act_dim, output_shape=1, **net)
    qf2 = networks.QNet(input_shape=obs_dim + act_dim, output_shape=1, **net)
    collector = BaseCollector(env=env, pf=pf, replay_buffer=replay_buffer, device=device,
                             train_render=False, **params["collector"])
    general = dict(params['general_setting'], env=collector.env, replay_buffer=replay_buffer, logger=logger, device=device,
                   collector=collector, save_dir=osp.join(logger.work_dir, "model"))
    TwinSACQ(pf=pf, qf1=qf1, qf2=qf2, **params["twin_sac_q"], **general).train()


if __name__ == "__main__":
    main()
