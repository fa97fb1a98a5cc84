"""Functional fp32 MLP / Gaussian-tanh policy oracle (torch CPU).

Restates, as pure functions over a flat parameter list, what the reference
builds from nn.Modules:

* ``mlp``: Linear -> act after *every* hidden layer (including the last one,
  torchrl/networks/base.py:22-44), then a linear head (nets.py:34-52);
* ``init_mlp``: hidden layers ``basic_init`` = U(+-sqrt(1/size[0])) with
  size[0] == out_features and bias 0.1 (networks/init.py:5-31); head
  ``uniform_init`` = U(+-3e-3) for weight and bias (init.py:34-36);
* ``gaussian_head``: state-independent ``logstd`` clamped to [-20, 2]
  (policies/continuous_policy.py:8-9, 180-188);
* ``tanh_normal_log_prob``: atanh(a) = log((1+a)/(1-a))/2, Normal log-density
  minus log(1 - a^2 + 1e-6) (policies/distribution.py:33-45);
* ``normal_entropy``: 0.5 + 0.5 log(2 pi) + log sigma summed over actions
  (distribution.py:78-79, continuous_policy.py:100,143).
"""
import math
import torch

LOG_SIG_MAX = 2.0
LOG_SIG_MIN = -20.0
ACTS = {"tanh": torch.tanh, "relu": torch.relu}


def init_mlp(in_dim, hidden, out_dim, generator=None):
    """Returns [W1, b1, ..., Wh, bh] with nn.Linear layout W=(out, in)."""
    params = []
    d = in_dim
    for h in hidden:
        bound = math.sqrt(1.0 / h)                      # fan_in := size[0] = out_features
        w = (torch.rand(h, d, generator=generator) * 2 - 1) * bound
        b = torch.full((h,), 0.1)
        params += [w, b]
        d = h
    w = (torch.rand(out_dim, d, generator=generator) * 2 - 1) * 3e-3
    b = (torch.rand(out_dim, generator=generator) * 2 - 1) * 3e-3
    params += [w, b]
    return [p.float() for p in params]


def mlp(x, params, act="tanh"):
    f = ACTS[act]
    n_layers = len(params) // 2
    for i in range(n_layers - 1):
        x = f(torch.nn.functional.linear(x, params[2 * i], params[2 * i + 1]))
    return torch.nn.functional.linear(x, params[-2], params[-1])


def gaussian_head(mean, logstd):
    logstd = torch.clamp(logstd, LOG_SIG_MIN, LOG_SIG_MAX)
    std = torch.exp(logstd)
    return mean, std.unsqueeze(0).expand_as(mean), logstd


def normal_log_density(z, mean, std):
    var = std * std
    return -((z - mean) ** 2) / (2 * var) - torch.log(std) - math.log(math.sqrt(2 * math.pi))


def tanh_normal_log_prob(action, mean, std, eps=1e-6):
    pre = torch.log((1 + action) / (1 - action)) / 2
    return normal_log_density(pre, mean, std) - torch.log(1 - action * action + eps)


def normal_entropy(std):
    return 0.5 + 0.5 * math.log(2 * math.pi) + torch.log(std)


def policy_update_terms(obs, acts, pf_params, logstd, act="tanh", tanh_action=True):
    """== pf.update(obs, acts) (continuous_policy.py:134-153): log_prob (B,1), ent (B,1)."""
    mean = mlp(obs, pf_params, act)
    mean, std, log_std = gaussian_head(mean, logstd)
    if tanh_action:
        lp = tanh_normal_log_prob(acts, mean, std)
    else:
        lp = normal_log_density(acts, mean, std)
    return {"mean": mean, "std": std, "log_std": log_std,
            "log_prob": lp.sum(-1, keepdim=True),
            "ent": normal_entropy(std).sum(-1, keepdim=True)}


def explore_action(obs, pf_params, logstd, noise, act="tanh", tanh_action=True):
    """== pf.explore(obs)['action'] with the N(0,1) draw supplied by the caller
    (distribution.py:60-76: z = mean + std * eps, action = tanh(z))."""
    mean = mlp(obs, pf_params, act)
    mean, std, _ = gaussian_head(mean, logstd)
    z = mean + std * noise
    return torch.tanh(z) if tanh_action else z
