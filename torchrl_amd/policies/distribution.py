"""TanhNormal with the reference's formulas (torchrl/policies/distribution.py:5-79):
X = tanh(Z), Z ~ N(mean, std); log_prob(x) uses atanh(x) = log((1+x)/(1-x))/2
when no pre-tanh value is given and subtracts log(1 - x^2 + eps); rsample draws
its N(0,1) on the CPU generator and moves it to the mean's device, so seeded
runs reproduce the reference's exploration-noise stream (its Q5)."""
import torch
from torch.distributions import Distribution, Normal


class TanhNormal(Distribution):
    def __init__(self, normal_mean, normal_std, epsilon=1e-6):
        self.normal_mean = normal_mean
        self.normal_std = normal_std
        self.normal = Normal(normal_mean, normal_std)
        self.epsilon = epsilon

    def _squash(self, z, return_pre):
        return (torch.tanh(z), z) if return_pre else torch.tanh(z)

    def sample_n(self, n, return_pre_tanh_value=False):
        return self._squash(self.normal.sample_n(n), return_pre_tanh_value)

    def log_prob(self, value, pre_tanh_value=None):
        if pre_tanh_value is None:
            pre_tanh_value = torch.log((1 + value) / (1 - value)) / 2
        return self.normal.log_prob(pre_tanh_value) - torch.log(1 - value * value + self.epsilon)

    def sample(self, return_pretanh_value=False):
        return self._squash(self.normal.sample().detach(), return_pretanh_value)

    def rsample(self, return_pretanh_value=False):
        unit = torch.randn(self.normal_mean.size())                      # CPU generator (distribution.py:67-70)
        z = self.normal_mean + self.normal_std * unit.to(self.normal_mean.device)
        return self._squash(z, return_pretanh_value)

    def entropy(self):
        return self.normal.entropy()
