"""PPO with the envs stepped in 4 worker processes -- the wiring of the reference's examples/ppo_continuous_vec_subproc.py
(`get_subprocvec_env(env_name, env_param, vec_env_nums, 4)`), on a host id:

    python examples/ppo_continuous_vec_subproc.py --config config/ppo_pendulum_host.json --vec_env_nums 16 --seed 0 --overwrite
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
from torchrl.env import get_subprocvec_env                    # noqa: E402
from torchrl.replay_buffers.on_policy import OnPolicyReplayBuffer  # noqa: E402
from torchrl.utils import Logger, get_args, get_params    # noqa: E402


def main():
    args = get_args()
    params = get_params(args.config)
    device = torch.device("cuda:{}".format(args.device) if args.cuda else "cpu")
    if device.type == "cuda":
        torch.cuda.set_device(device)                    # envs / replay buffers allocate on the current device

    env = get_subprocvec_env(params["env_name"], params["env"], args.vec_env_nums, 4)
    eval_env = get_subprocvec_env(params["env_name"], params["env"], args.vec_env_nums, 4)
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
    general = dict(params['general_setting'], env=collector.env, replay_buffer=replay_buffer, logger=logger,
                   device=device, collector=collector, save_dir=osp.join(logger.work_dir, "model"))
    try:
        PPO(pf=pf, vf=vf, **params["ppo"], **general).train()
    finally:
        collector.terminate()                                  # stops the worker processes


if __name__ == "__main__":
    main()
