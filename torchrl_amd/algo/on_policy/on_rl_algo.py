"""On-policy epoch post-processing (torchrl/algo/on_policy/on_rl_algo.py:6-47):
last_value = vf(next_obs[T-1]) * (1 - terminals[T-1]) then GAE or discounted
return over the whole buffer -- vf forward and the scan both run as HIP kernels,
the terminal mask is folded into the scan kernel."""
import torch

from ..rl_algo import RLAlgo


class OnRLAlgo(RLAlgo):
    def __init__(self, shuffle=True, tau=None, gae=True, **kwargs):
        super().__init__(**kwargs)
        self.sample_key = ["obs", "acts", "advs", "estimate_returns"]
        self.shuffle = shuffle
        self.tau = tau
        self.gae = gae

    def process_epoch_samples(self):
        sample = self.replay_buffer.last_sample(['next_obs', 'terminals', "time_limits"])
        with torch.no_grad():
            last_value = self.vf(sample['next_obs'].to(self.device))
        if self.gae:
            self.replay_buffer.generalized_advantage_estimation(
                last_value, self.discount, self.tau, last_terminal=sample["terminals"])
        else:
            self.replay_buffer.discount_reward(last_value, self.discount, last_terminal=sample["terminals"])

    def update_per_epoch(self):
        self.process_epoch_samples()
        for batch in self.replay_buffer.one_iteration(self.batch_size, self.sample_key, self.shuffle):
            infos = self.update(batch)
            self.logger.add_update_info(infos)

    @property
    def networks(self):
        return [self.pf, self.vf]
