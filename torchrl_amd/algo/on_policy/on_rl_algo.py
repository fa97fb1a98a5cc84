"""On-policy epoch post-processing (torchrl/algo/on_policy/on_rl_algo.py:6-47).

After a rollout the ring holds T rows x N envs.  `process_epoch_samples` turns the stored rewards / values into
advantages and return estimates: V(next_obs of the last row), masked by that row's terminals inside the scan kernel,
bootstraps either the GAE scan (`gae=True`, lambda = `tau`) or the plain discounted return; both the value forward and
the reverse scan are HIP kernels (k_ppo.hip / k_gae.hip), nothing leaves the device.  `update_per_epoch` is the reference's
single pass over `one_iteration` minibatches; PPO / A2C / TRPO / V-MPO override it with their own loops."""
import torch

from ..rl_algo import RLAlgo

_LAST_ROW_KEYS = ("next_obs", "terminals", "time_limits")


class OnRLAlgo(RLAlgo):
    sample_key = ("obs", "acts", "advs", "estimate_returns")

    def __init__(self, shuffle=True, tau=None, gae=True, **kwargs):
        super().__init__(**kwargs)
        self.sample_key = list(type(self).sample_key)
        self.shuffle, self.tau, self.gae = shuffle, tau, gae

    @property
    def networks(self):
        return [self.pf, self.vf]

    def process_epoch_samples(self):
        buf = self.replay_buffer
        last = buf.last_sample(list(_LAST_ROW_KEYS))
        if getattr(buf, "_boot_fresh", False):                           # the fused rollout's value pass left V(next_obs) of
            bootstrap = buf._boot                                        # the last row (same parameters: nothing stepped since)
        else:
            with torch.no_grad():
                bootstrap = self.vf(last["next_obs"].to(self.device))
        if not self.gae:
            return buf.discount_reward(bootstrap, self.discount, last_terminal=last["terminals"])
        return buf.generalized_advantage_estimation(bootstrap, self.discount, self.tau, last_terminal=last["terminals"])

    def update_per_epoch(self):
        self.process_epoch_samples()
        for minibatch in self.replay_buffer.one_iteration(self.batch_size, self.sample_key, self.shuffle):
            self.logger.add_update_info(self.update(minibatch))
