"""Gaussian policies with the reference's API
(torchrl/policies/continuous_policy.py:77-188).

`GuassianContPolicyBasicBias` (PPO/A2C: MLP mean + state-independent `logstd`
parameter initialised to log(log_init), clamped to [-20, 2]) is the policy of
the benchmark path.  Inside the fused collector / PPO kernels its parameters are
read from `flat_params()` (MLP2 block + logstd tail); the methods below keep the
reference protocol (`explore` / `update` / `eval_act` dicts) for callers that
use the policy directly: the mean comes from the HIP MLP kernel, log-probs from
trl_gauss_logp_f32.
"""
import numpy as np
import torch
import torch.nn as nn
from torch.distributions import Normal

from .. import _C
from .. import networks
from .distribution import TanhNormal

LOG_SIG_MAX = 2
LOG_SIG_MIN = -20


class GuassianContPolicyBase:
    def eval_act(self, x):
        return self.torch_eval_act(x).squeeze(0).cpu().numpy()

    def torch_eval_act(self, x):
        with torch.no_grad():
            mean, _, _ = self.forward(x)
        return (torch.tanh(mean) if self.tanh_action else mean).detach()

    def _dist(self, mean, std):
        return TanhNormal(mean, std) if self.tanh_action else Normal(mean, std)

    def explore(self, x, return_log_probs=False, return_pre_tanh=False):
        """The reference's direct-call protocol (continuous_policy.py:92-131) on torch.distributions; the collectors do
        not come through here (their sampling is in the rollout / rsample kernels)."""
        mean, std, log_std = self.forward(x)
        if mean.is_cuda:
            _C.note_eager(type(self).__name__ + ".explore", "torch.distributions sampling")
        dis = self._dist(mean, std)
        out = {"mean": mean, "log_std": log_std, "std": std,
               "ent": dis.entropy().sum(-1, keepdim=True)}
        if return_log_probs:
            if self.tanh_action:
                action, z = dis.rsample(return_pretanh_value=True)
                log_prob = dis.log_prob(action, pre_tanh_value=z)
                out["pre_tanh"] = z.squeeze(0)
            else:
                action = dis.sample()
                log_prob = dis.log_prob(action)
            out["log_prob"] = log_prob.sum(dim=-1, keepdim=True)
        elif self.tanh_action:
            if return_pre_tanh:
                action, z = dis.rsample(return_pretanh_value=True)
                out["pre_tanh"] = z.squeeze(0)
            action = dis.rsample(return_pretanh_value=False)
        else:
            action = dis.sample()
        out["action"] = action.squeeze(0)
        return out

    def update(self, obs, actions):
        mean, std, log_std = self.forward(obs)
        if mean.is_cuda and not (torch.is_grad_enabled() and mean.requires_grad):
            ls = log_std if log_std.dim() == 1 else None
            if ls is not None:
                lp = _C.gauss_logp(mean.contiguous(), actions.float().contiguous(), ls.float().contiguous(),
                                   self.tanh_action).unsqueeze(-1)
            else:
                _C.note_eager(type(self).__name__ + ".update", "state-dependent std has no log-prob kernel")
                lp = self._dist(mean, std).log_prob(actions).sum(-1, keepdim=True)
        else:
            if mean.is_cuda:
                _C.note_eager(type(self).__name__ + ".update", "autograd is on")
            lp = self._dist(mean, std).log_prob(actions).sum(-1, keepdim=True)
        return {"mean": mean, "dis": Normal(mean, std), "log_std": log_std, "std": std,
                "log_prob": lp, "ent": Normal(mean, std).entropy().sum(-1, keepdim=True)}


class GuassianContPolicy(networks.Net, GuassianContPolicyBase):
    """State-dependent std (SAC): head emits [mean | log_std] (continuous_policy.py:156-170)."""

    def __init__(self, tanh_action=False, **kwargs):
        super().__init__(**kwargs)
        self.continuous = True
        self.tanh_action = tanh_action

    def forward(self, x):
        mean, log_std = super().forward(x).chunk(2, dim=-1)
        log_std = torch.clamp(log_std, LOG_SIG_MIN, LOG_SIG_MAX)
        return mean, torch.exp(log_std), log_std


class GuassianContPolicyBasicBias(networks.Net, GuassianContPolicyBase):
    def __init__(self, output_shape, tanh_action=False, log_init=0.125, **kwargs):
        super().__init__(output_shape=output_shape, **kwargs)
        self.continuous = True
        self.logstd = nn.Parameter(torch.ones(output_shape) * np.log(log_init))
        self.tanh_action = tanh_action

    def _extra_flat_params(self):
        return [self.logstd]

    def forward(self, x):
        mean = super().forward(x)
        logstd = torch.clamp(self.logstd, LOG_SIG_MIN, LOG_SIG_MAX)
        std = torch.exp(logstd).unsqueeze(0).expand_as(mean)
        return mean, std, logstd


class DetContPolicy(networks.Net):
    """Deterministic policy (torchrl/policies/continuous_policy.py:28-47): action = [tanh](mlp(x))."""

    def __init__(self, tanh_action=False, **kwargs):
        super().__init__(**kwargs)
        self.continuous = True
        self.tanh_action = tanh_action

    def forward(self, x):
        out = super().forward(x)
        return torch.tanh(out) if self.tanh_action else out

    def eval_act(self, x):
        with torch.no_grad():
            return self.forward(x).squeeze(0).detach().cpu().numpy()

    def explore(self, x):
        return {"action": self.forward(x).squeeze(0)}


class FixGuassianContPolicy(networks.Net):
    def __init__(self, norm_std_explore, tanh_action=False, **kwargs):
        super().__init__(**kwargs)
        self.continuous = True
        self.tanh_action = tanh_action
        self.norm_std_explore = norm_std_explore

    def forward(self, x):
        out = super().forward(x)
        return torch.tanh(out) if self.tanh_action else out

    def eval_act(self, x):
        with torch.no_grad():
            return self.forward(x).squeeze(0).detach().cpu().numpy()

    def explore(self, x):
        action = self.forward(x).squeeze(0)
        noise = Normal(0, self.norm_std_explore).sample(action.shape).to(action.device)
        return {"action": action + noise}


class UniformPolicyContinuous(nn.Module):
    def __init__(self, action_shape):
        super().__init__()
        self.continuous = True
        self.action_shape = action_shape

    def forward(self, x):
        return torch.Tensor(np.random.uniform(-1., 1., self.action_shape))

    def explore(self, x):
        return {"action": self.forward(x)}
