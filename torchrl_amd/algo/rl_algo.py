"""Epoch driver with the reference's constructor and `train()` protocol (torchrl/algo/rl_algo.py:14-195).

One epoch = collect (`collector.train_one_epoch`) -> `update_per_epoch` -> every `eval_interval` epochs evaluate, keep a
`best` snapshot and emit one logger row -> every `save_interval` epochs a numbered snapshot; a `finish` snapshot at
the end.  Snapshot files keep the reference's names: `model_{name}_{epoch}.pth` per entry of `snapshot_networks` and
`_obs_normalizer_{epoch}.pkl` when the env carries a running normaliser.  The logger row has the reference's keys
(including the padded `Train___Time` / `Eval____Time`); the two accumulated timers restart after every logged row."""
import os
import pickle
import time
from collections import deque
from contextlib import contextmanager

import gym
import numpy as np
import torch

from . import utils as atu

_RECENT = 30                       # episodes in the running reward averages


def _mean_or_nan(values):
    return float("nan") if len(values) == 0 else np.mean(values)


class RLAlgo:
    def __init__(self, env=None, replay_buffer=None, collector=None, logger=None, grad_clip=None,
                 discount=0.99, num_epochs=3000, batch_size=128, device='cpu', save_interval=100,
                 eval_interval=1, save_dir=None):
        self.env, self.replay_buffer, self.collector, self.logger = env, replay_buffer, collector, logger
        self.continuous = isinstance(env.action_space, gym.spaces.Box)
        self.device = torch.device(device)
        self.grad_clip, self.discount, self.batch_size = grad_clip, discount, batch_size
        self.num_epochs, self.eval_interval, self.save_interval = num_epochs, eval_interval, save_interval
        self.epoch_frames = collector.epoch_frames
        self.sample_key = None
        self.training_update_num = 0
        self.current_epoch = 0
        self.best_eval = None
        self.episode_rewards = deque(maxlen=_RECENT)
        self.training_episode_rewards = deque(maxlen=_RECENT)
        self.explore_time = self.train_time = 0
        self.start = time.time()
        self.save_dir = save_dir
        if save_dir is not None:
            os.makedirs(save_dir, exist_ok=True)

    # ---- what an algorithm overrides ----
    networks = property(lambda self: [])
    snapshot_networks = property(lambda self: [])
    target_networks = property(lambda self: [])

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

    def to(self, device):
        for net in self.networks:
            net.to(device)

    # ---- snapshots ----
    def snapshot(self, prefix, epoch):
        """rl_algo.py:83-94.  With one process per GPU every rank holds identical parameters: rank 0 writes.  Parameters
        are views of one flat buffer here, so each tensor is cloned -- a .pth holds that network only."""
        from .. import dist
        if prefix is None or dist.rank() != 0:                            # (rank of THIS package's process group: a process that
            return                                                     # merely inherited RANK != 0 still writes its files)
        dist.check_comm()                                              # never snapshot parameters stepped with a partial gradient sum
        from ..networks import nets as _nets
        for _name, network in self.snapshot_networks:                  # ... nor parameters another stream is still stepping
            _nets.settle(network)
        normalizer = getattr(self.env, "_obs_normalizer", None)
        if normalizer is not None:
            with open(os.path.join(prefix, "_obs_normalizer_%s.pkl" % (epoch,)), "wb") as handle:
                pickle.dump(normalizer, handle)
        for name, network in self.snapshot_networks:
            state = {k: v.detach().clone() for k, v in network.state_dict().items()}
            torch.save(state, os.path.join(prefix, "model_%s_%s.pth" % (name, epoch)))

    # ---- the loop ----
    @contextmanager
    def _timed(self, attr):
        t0 = time.time()
        yield
        setattr(self, attr, getattr(self, attr) + time.time() - t0)

    def _evaluate_and_log(self, epoch, total_frames, collected, extra):
        t0 = time.time()
        result = self.collector.eval_one_epoch()
        eval_time = time.time() - t0
        rewards = result.pop("eval_rewards")
        self.episode_rewards.extend(rewards)
        score = np.mean(rewards)
        if self.best_eval is None or score > self.best_eval:
            self.best_eval = score
            self.snapshot(self.save_dir, 'best')
        row = {"Running_Average_Rewards": np.mean(self.episode_rewards),
               "Train_Epoch_Reward": collected["train_epoch_reward"],
               "Running_Training_Average_Rewards": _mean_or_nan(self.training_episode_rewards),
               "Explore_Time": self.explore_time, "Train___Time": self.train_time, "Eval____Time": eval_time}
        row.update(result)
        row.update(extra)
        self.explore_time = self.train_time = 0
        self.logger.add_epoch_info(epoch, total_frames, time.time() - self.start, row)
        self.start = time.time()

    def train(self):
        # This loop is the reference's (rl_algo.py:96-164): between two rollouts nothing draws from the CPU torch generator
        # (updates and greedy evaluation are deterministic given the batch), so an on-policy collector in the reference's
        # noise mode may draw the next rollout's exploration noise while the device is busy (TRL_PREFETCH_NOISE=0: in place).
        # The worker draws from private generators; the default generator is only read / set on this thread when a block is
        # taken, and a block is only used if the generator still is where the block started -- a callback or a custom policy
        # that draws from (or re-seeds) the CPU generator between two rollouts just makes that block be drawn in place, in
        # the reference's order (collector/on_policy.py::_NoisePrefetcher, logged once).
        if hasattr(self.collector, "prefetch_noise") and os.environ.get("TRL_PREFETCH_NOISE") != "0":
            self.collector.prefetch_noise = True
        self.pretrain()
        total_frames = getattr(self, "pretrain_frames", 0)
        self.start_epoch()
        for epoch in range(self.num_epochs):
            self.current_epoch = epoch
            self.start_epoch()
            with self._timed("explore_time"):
                collected = self.collector.train_one_epoch()                 # device collectors: read back on first access
            with self._timed("train_time"):
                self.update_per_epoch()
            # after the update has been launched: the collector's read-back then overlaps it instead of idling the GPU
            # (the two timers measure host time; on the device path the rollout's wait shows up under Train___Time)
            self.training_episode_rewards.extend(collected["train_rewards"])
            extra = self.finish_epoch()
            total_frames += self.epoch_frames
            if epoch % self.eval_interval == 0:
                self._evaluate_and_log(epoch, total_frames, collected, extra)
            if epoch % self.save_interval == 0:
                self.snapshot(self.save_dir, epoch)
        self.snapshot(self.save_dir, "finish")
        self.collector.terminate()
        self.logger.finish()

    def _update_target_networks(self):
        """Polyak step every update, or a hard copy every `target_hard_update_period` updates (rl_algo.py:169-176)."""
        if self.use_soft_update:
            move = lambda net, target: atu.soft_update_from_to(net, target, self.tau)
        elif self.training_update_num % self.target_hard_update_period == 0:
            move = atu.copy_model_params_from_to
        else:
            return
        for net, target in self.target_networks:
            move(net, target)
