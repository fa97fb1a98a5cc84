"""Off-policy epoch driver (torchrl/algo/off_policy/off_rl_algo.py:8-84).

Per epoch: `opt_times` x {uniform replay sample -> `update` -> log the info dict}.  `pretrain()` runs
`pretrain_epochs` collection-only epochs (logged without a csv row) and accounts their frames in `pretrain_frames`.
Engines that replay a captured graph expose `static_batch()`: the replay gather then writes into those fixed-address
tensors (`random_batch(..., out=)`, an argument the reference does not have)."""
import time

import numpy as np

from ..rl_algo import RLAlgo


class OffRLAlgo(RLAlgo):
    sample_key = ("obs", "next_obs", "acts", "rewards", "terminals")

    def __init__(self, pretrain_epochs=0, min_pool=0, target_hard_update_period=1000,
                 use_soft_update=True, tau=0.001, opt_times=1, **kwargs):
        super().__init__(**kwargs)
        self.sample_key = list(type(self).sample_key)
        self.pretrain_epochs, self.min_pool, self.opt_times = pretrain_epochs, min_pool, opt_times
        self.use_soft_update, self.tau = use_soft_update, tau
        self.target_hard_update_period = target_hard_update_period

    def _one_update(self):
        self.logger.add_update_info(self.update(self._sample()))

    def _sample(self):
        extra = {}
        static = getattr(self, "static_batch", None)
        out = static() if static is not None else None                  # (None until the engine has seen its first batch)
        if out is not None:
            extra["out"] = out
        return self.replay_buffer.random_batch(self.batch_size, self.sample_key, **extra)

    def update_per_epoch(self):
        deferred = getattr(self, "update_deferred", None)
        if deferred is None:
            for _ in range(self.opt_times):
                self._one_update()
        else:
            # sample -> update x opt_times launched back to back (the host never waits inside the loop, so the next
            # sample's index upload and launches overlap the running update); the info dicts reach the logger in the
            # reference's order after one read-back
            # (an engine with `update_epoch_deferred` takes all `opt_times` samples + updates as one replayed graph when
            # its conditions hold -- the index sets are drawn here on the host in the same order -- else returns None)
            whole = getattr(self, "update_epoch_deferred", None)
            pending = whole(self.opt_times) if whole is not None else None
            if pending is None:
                pending = [deferred(self._sample()) for _ in range(self.opt_times)]
            later = getattr(self.logger, "add_update_infos_later", None)
            if later is not None and not getattr(self, "eager_update_infos", False):
                later(lambda: self.resolve_updates(pending))             # read when the logger writes its next row: the
            else:                                                        # next epoch's collection is launched meanwhile
                for info in self.resolve_updates(pending):
                    self.logger.add_update_info(info)
        check = getattr(self.replay_buffer, "check_overrun", None)       # frame-dedup replay: a batch that asked for an
        if check is not None:                                            # overwritten frame fails the epoch, loudly
            check()

    def update_per_timestep(self):
        if self.replay_buffer.num_steps_can_sample() > max(self.min_pool, self.batch_size):
            self.update_per_epoch()

    def pretrain(self):
        self.pretrain_frames = self.pretrain_epochs * self.epoch_frames
        for index in range(self.pretrain_epochs):
            began = time.time()
            self.start_epoch()
            collected = self.collector.train_one_epoch()
            self.training_episode_rewards.extend(collected["train_rewards"])
            recent = self.training_episode_rewards
            row = {"Train_Epoch_Reward": collected["train_epoch_reward"],
                   "Running_Training_Average_Rewards": np.mean(recent) if len(recent) else float("nan")}
            row.update(self.finish_epoch())
            self.logger.add_epoch_info(index, (index + 1) * self.epoch_frames, time.time() - began, row, csv_write=False)
        self.logger.log("Finished Pretrain")
