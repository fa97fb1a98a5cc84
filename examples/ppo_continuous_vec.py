"""PPO on a vectorised synthetic env -- the wiring of the reference's
examples/ppo_continuous_vec.py (same imports, same constructor kwargs, same JSON
sections), pointed at the on-GPU env id:

    python examples/ppo_continuous_vec.py --config config/ppo_synth_halfcheetah.json \
        --vec_env_nums 2048 --seed 0 --overwrite
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
from torchrl.collector.on_policy import VecOnPolicyCollector  # noqa: E402
from torchrl.env import get_vec_env                       # noqa: E402
from torchrl.replay_buffers.on_policy import OnPolicyReplayBuffer  # noqa: E402
from torchrl.utils import Logger, get_args, get_params    # noqa: E402


def main():
    args = get_args()
    params = get_params(args.config)
    device = torch.device("cuda:{}".format(args.device) if args.cuda else "cpu")
    if device.type == "cuda":
        torch.cuda.set_device(device)                    # envs / replay buffers allocate on the current device

    env = get_vec_env(params["env_name"], params["env"], args.vec_env_nums)
    eval_env = get_vec_env(params["env_name"], params["env"], args.vec_env_nums)
    env.seed(args.seed)
    for seed_fn in (torch.manual_seed, np.random.seed, random.seed):
        seed_fn(args.seed)

    name = args.id if args.id is not None else osp.splitext(osp.basename(args.config))[0]
    logger = Logger(name, params['env_name'], args.seed, params, args.log_dir, args.overwrite)

    replay_buffer = OnPolicyReplayBuffer(env_nums=args.vec_env_nums,
                                         max_replay_buffer_size=int(params['replay_buffer']['size']),
                                         time_limit_filter=params['replay_buffer']['time_limit_filter'])
    net = dict(params['net'], base_type=networks.MLPBase, activation_func=torch.nn.Tanh)
    pf = policies.GuassianContPolicyBasicBias(input_shape=env.observation_space.shape[0],
                                              output_shape=env.action_space.shape[0], **net, **params['policy'])
    vf = networks.Net(input_shape=env.observation_space.shape, output_shape=1, **net)
    collector = VecOnPolicyCollector(vf, env=env, eval_env=eval_env, pf=pf, replay_buffer=replay_buffer,
                                     device=device, train_render=False, **params["collector"])
    general = dict(params['general_setting'], env=env, replay_buffer=replay_buffer, logger=logger,
                   device=device, collector=collector, save_dir=osp.join(logger.work_dir, "model"))
    PPO(pf=pf, vf=vf, **params["ppo"], **general).train()


if __name__ == "__main__":
    main()
