"""PPO on ONE synthetic env -- the wiring of the reference's examples/ppo_continuous.py (`get_env`,
`OnPolicyCollectorBase`, a replay buffer without `env_nums`) and the hyper-parameters of its
config/ppo_halfcheetah.json (SURVEY.md 8(d) cfg 1: T = 2048, batch 64, 10 opt epochs, obs_norm):

    python examples/ppo_continuous.py --config config/ppo_synth_halfcheetah_single.json --seed 0 --overwrite
"""
import os.path as osp
import random
import sys

import numpy as np
import torch

sys.path.append(osp.join(osp.dirname(osp.abspath(__file__)), ".."))
import torchrl.networks as networks                       # noqa: E402
import torchrl.policies as policies                       # noqa: E402
from torchrl.algo import PPO                              # noqa: E402
from torchrl.collector.on_policy import OnPolicyCollectorBase  # noqa: E402
from torchrl.env import get_env                           # noqa: E402
from torchrl.replay_buffers.on_policy import OnPolicyReplayBuffer  # noqa: E402
from torchrl.utils import Logger, get_args, get_params    # noqa: E402


def main():
    args = get_args()
    params = get_params(args.config)
    device = torch.device("cuda:{}".format(args.device) if args.cuda else "cpu")
    if device.type == "cuda":
        torch.cuda.set_device(device)                    # envs / replay buffers allocate on the current device

    env = get_env(params['env_name'], params['env'])
    env.train()
    eval_env = get_env(params['env_name'], params['env'])
    eval_env.eval()
    env.seed(args.seed)
    for seed_fn in (torch.manual_seed, np.random.seed, random.seed):
        seed_fn(args.seed)

    name = args.id if args.id is not None else osp.splitext(osp.basename(args.config))[0]
    logger = Logger(name, params['env_name'], args.seed, params, args.log_dir, overwrite=args.overwrite)

    replay_buffer = OnPolicyReplayBuffer(int(params['replay_buffer']['size']),
                                         time_limit_filter=params['replay_buffer']['time_limit_filter'])
    net = dict(params['net'], base_type=networks.MLPBase, activation_func=torch.nn.Tanh)
    pf = policies.GuassianContPolicyBasicBias(input_shape=env.observation_space.shape[0],
                                              output_shape=env.action_space.shape[0], **net, **params['policy'])
    vf = networks.Net(input_shape=env.observation_space.shape, output_shape=1, **net)
    collector = OnPolicyCollectorBase(vf, env=env, eval_env=eval_env, pf=pf, replay_buffer=replay_buffer,
                                      device=device, train_render=False, **params["collector"])
    general = dict(params['general_setting'], env=env, replay_buffer=replay_buffer, logger=logger,
                   device=device, collector=collector, save_dir=osp.join(logger.work_dir, "model"))
    PPO(pf=pf, vf=vf, **params["ppo"], **general).train()


if __name__ == "__main__":
    main()
