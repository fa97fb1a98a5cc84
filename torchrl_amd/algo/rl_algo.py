"""Epoch driver with the reference's constructor and `train()` protocol
(torchrl/algo/rl_algo.py:14-195): collect one epoch, update, evaluate every
`eval_interval`, snapshot on best / every `save_interval` / at the end with the
reference's file names (`model_{name}_{epoch}.pth`)."""
import os.path as osp
import pathlib
import pickle
import time
from collections import deque

import gym
import numpy as np
import torch

from . import utils as atu


class RLAlgo:
    def __init__(self, env=None, replay_buffer=None, collector=None, logger=None, grad_clip=None,
                 discount=0.99, num_epochs=3000, batch_size=128, device='cpu', save_interval=100,
                 eval_interval=1, save_dir=None):
        self.env = env
        self.continuous = isinstance(self.env.action_space, gym.spaces.Box)
        self.replay_buffer = replay_buffer
        self.collector = collector
        self.device = torch.device(device)
        self.discount = discount
        self.num_epochs = num_epochs
        self.epoch_frames = self.collector.epoch_frames
        self.batch_size = batch_size
        self.training_update_num = 0
        self.sample_key = None
        self.grad_clip = grad_clip
        self.logger = logger
        self.episode_rewards = deque(maxlen=30)
        self.training_episode_rewards = deque(maxlen=30)
        self.save_interval = save_interval
        self.save_dir = save_dir
        if self.save_dir is not None:
            pathlib.Path(self.save_dir).mkdir(parents=True, exist_ok=True)
        self.best_eval = None
        self.eval_interval = eval_interval
        self.explore_time = 0
        self.train_time = 0
        self.start = time.time()
        self.current_epoch = 0

    # hooks
    def start_epoch(self):
        pass

    def finish_epoch(self):
        return {}

    def pretrain(self):
        pass

    def update_per_epoch(self):
        pass

    def update(self, batch):
        raise NotImplementedError

    def snapshot(self, prefix, epoch):
        if prefix is None:
            return
        normalizer = getattr(self.env, "_obs_normalizer", None)
        if normalizer is not None:
            with open(osp.join(prefix, "_obs_normalizer_{}.pkl".format(epoch)), "wb") as f:
                pickle.dump(normalizer, f)
        for name, network in self.snapshot_networks:
            torch.save(network.state_dict(), osp.join(prefix, "model_{}_{}.pth".format(name, epoch)))

    def train(self):
        self.pretrain()
        total_frames = getattr(self, "pretrain_frames", 0)
        self.start_epoch()
        for epoch in range(self.num_epochs):
            self.current_epoch = epoch
            self.start_epoch()

            t0 = time.time()
            epoch_info = self.collector.train_one_epoch()
            self.training_episode_rewards.extend(epoch_info["train_rewards"])
            self.explore_time += time.time() - t0

            t0 = time.time()
            self.update_per_epoch()
            self.train_time += time.time() - t0

            finish_info = self.finish_epoch()
            total_frames += self.epoch_frames

            if epoch % self.eval_interval == 0:
                t0 = time.time()
                eval_infos = self.collector.eval_one_epoch()
                eval_time = time.time() - t0
                self.episode_rewards.extend(eval_infos["eval_rewards"])
                mean_eval = np.mean(eval_infos["eval_rewards"])
                if self.best_eval is None or mean_eval > self.best_eval:
                    self.best_eval = mean_eval
                    self.snapshot(self.save_dir, 'best')
                del eval_infos["eval_rewards"]
                infos = {
                    "Running_Average_Rewards": np.mean(self.episode_rewards),
                    "Train_Epoch_Reward": epoch_info["train_epoch_reward"],
                    "Running_Training_Average_Rewards":
                        np.mean(self.training_episode_rewards) if len(self.training_episode_rewards) else float("nan"),
                    "Explore_Time": self.explore_time,
                    "Train___Time": self.train_time,
                    "Eval____Time": eval_time,
                }
                self.explore_time = 0
                self.train_time = 0
                infos.update(eval_infos)
                infos.update(finish_info)
                self.logger.add_epoch_info(epoch, total_frames, time.time() - self.start, infos)
                self.start = time.time()

            if epoch % self.save_interval == 0:
                self.snapshot(self.save_dir, epoch)

        self.snapshot(self.save_dir, "finish")
        self.collector.terminate()
        self.logger.finish()

    def _update_target_networks(self):
        if self.use_soft_update:
            for net, target_net in self.target_networks:
                atu.soft_update_from_to(net, target_net, self.tau)
        elif self.training_update_num % self.target_hard_update_period == 0:
            for net, target_net in self.target_networks:
                atu.copy_model_params_from_to(net, target_net)

    @property
    def networks(self):
        return []

    @property
    def snapshot_networks(self):
        return []

    @property
    def target_networks(self):
        return []

    def to(self, device):
        for net in self.networks:
            net.to(device)
