"""Off-policy epoch driver (torchrl/algo/off_policy/off_rl_algo.py:8-84): `opt_times` x
{uniform replay sample -> update -> log} per epoch, collection-only pretrain epochs."""
import time

import numpy as np

from ..rl_algo import RLAlgo


class OffRLAlgo(RLAlgo):
    def __init__(self, pretrain_epochs=0, min_pool=0, target_hard_update_period=1000,
                 use_soft_update=True, tau=0.001, opt_times=1, **kwargs):
        super().__init__(**kwargs)
        self.pretrain_epochs = pretrain_epochs
        self.target_hard_update_period = target_hard_update_period
        self.use_soft_update = use_soft_update
        self.tau = tau
        self.opt_times = opt_times
        self.min_pool = min_pool
        self.sample_key = ["obs", "next_obs", "acts", "rewards", "terminals"]

    def _sample_and_update(self):
        out = self.static_batch() if hasattr(self, "static_batch") else None     # fixed-address inputs (graph replay)
        batch = (self.replay_buffer.random_batch(self.batch_size, self.sample_key, out=out) if out is not None
                 else self.replay_buffer.random_batch(self.batch_size, self.sample_key))
        self.logger.add_update_info(self.update(batch))

    def update_per_timestep(self):
        if self.replay_buffer.num_steps_can_sample() > max(self.min_pool, self.batch_size):
            for _ in range(self.opt_times):
                self._sample_and_update()

    def update_per_epoch(self):
        for _ in range(self.opt_times):
            self._sample_and_update()

    def pretrain(self):
        total_frames = 0
        self.pretrain_frames = self.pretrain_epochs * self.epoch_frames
        for pretrain_epoch in range(self.pretrain_epochs):
            start = time.time()
            self.start_epoch()
            epoch_info = self.collector.train_one_epoch()
            self.training_episode_rewards.extend(epoch_info["train_rewards"])
            finish_info = self.finish_epoch()
            total_frames += self.epoch_frames
            infos = {"Train_Epoch_Reward": epoch_info["train_epoch_reward"],
                     "Running_Training_Average_Rewards":
                         np.mean(self.training_episode_rewards) if len(self.training_episode_rewards) else float("nan")}
            infos.update(finish_info)
            self.logger.add_epoch_info(pretrain_epoch, total_frames, time.time() - start, infos, csv_write=False)
        self.logger.log("Finished Pretrain")
