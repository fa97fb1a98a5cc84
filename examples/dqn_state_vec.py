"""DQN on state-vector envs -- the wiring of the reference's examples/dqn_state_vec.py (MLP Q-network, epsilon-greedy
policy, VecCollector, BaseReplayBuffer) with its config/dqn_cartpole.json hyper-parameters, on a pure-Python cart-pole
behind `torchrl.env.VecEnv` (no gym in the image): env physics on the host, Q-network forward / backward, TD loss, Adam
and the target update on the GPU kernels.

    python examples/dqn_state_vec.py --config config/dqn_cartpole_host.json --vec_env_nums 8 --seed 0 --overwrite
"""
import os.path as osp
import random
import sys

import numpy as np
import torch

sys.path.append(osp.join(osp.dirname(osp.abspath(__file__)), ".."))
import torchrl.networks as networks                       # noqa: E402
import torchrl.policies as policies                       # noqa: E402
from torchrl.algo import DQN                              # noqa: E402
from torchrl.collector import VecCollector                # noqa: E402
from torchrl.env import VecEnv                            # noqa: E402
from torchrl.env.py_envs import CartPoleEnv               # noqa: E402
from torchrl.replay_buffers import BaseReplayBuffer       # noqa: E402
from torchrl.utils import Logger, get_args, get_params    # noqa: E402


def main():
    args = get_args()
    params = get_params(args.config)
    device = torch.device("cuda:{}".format(args.device))
    n = args.vec_env_nums
    env, eval_env = VecEnv(n, CartPoleEnv, ()), VecEnv(n, CartPoleEnv, ())
    env.seed(args.seed)
    eval_env.seed(args.seed + 1)
    for seed_fn in (torch.manual_seed, np.random.seed, random.seed):
        seed_fn(args.seed)
    name = args.id if args.id is not None else osp.splitext(osp.basename(args.config))[0]
    logger = Logger(name, params['env_name'], args.seed, params, args.log_dir, args.overwrite)
    replay_buffer = BaseReplayBuffer(env_nums=n, max_replay_buffer_size=int(params['replay_buffer']['size']),
                                     time_limit_filter=params['replay_buffer']['time_limit_filter'])
    net = dict(params['net'], base_type=networks.MLPBase, activation_func=torch.nn.ReLU)
    qf = networks.Net(input_shape=env.observation_space.shape, output_shape=env.action_space.n, **net)
    pf = policies.EpsilonGreedyDQNDiscretePolicy(qf, action_shape=env.action_space.n, **params['policy'])
    collector = VecCollector(env=env, pf=pf, eval_env=eval_env, replay_buffer=replay_buffer, device=device,
                             train_render=False, **params["collector"])
    general = dict(params['general_setting'], env=collector.env, replay_buffer=replay_buffer, logger=logger, device=device,
                   collector=collector, save_dir=osp.join(logger.work_dir, "model"))
    DQN(pf=pf, qf=qf, **params["dqn"], **general).train()


if __name__ == "__main__":
    main()
