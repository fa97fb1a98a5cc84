"""Twin-Q SAC update, torch-CPU fp32 oracle.

Restates torchrl/algo/off_policy/twin_sac_q.py:84-220 (+ continuous_policy.py:92-132,
162-170; distribution.py:33-76; algo/utils.py:16-20) over flat parameter lists:

  new_a, logp = rsample(pf(obs), eps1)            # CPU N(0,1) draw #1 (its Q5)
  alpha step: loss = -mean(log_alpha * (logp + H_target)); Adam; alpha = exp(log_alpha) AFTER it
  a', logp' = rsample(pf(next_obs), eps2)         # draw #2, no grad
  q_target = r + (1 - d) gamma (min(tq1, tq2)(next_obs, a') - alpha logp')
  qf_i loss = MSE(q_i(obs, act), q_target)
  policy loss = mean(alpha logp - min(q1, q2)(obs, new_a)) + w_std mean(log_std^2) + w_mean mean(mean^2)
  steps: pf, qf1, qf2 (each Adam, default eps 1e-8, optional clip_grad_norm_), then Polyak(tau).
"""

import numpy as np
import torch

from . import nets
from .ppo import AdamState, clip_global_norm


def rsample(head, eps, tanh_action=True):
    mean, log_std = head.chunk(2, dim=-1)
    log_std = torch.clamp(log_std, nets.LOG_SIG_MIN, nets.LOG_SIG_MAX)
    std = torch.exp(log_std)
    z = mean + std * eps
    lp = nets.normal_log_density(z, mean, std)
    if tanh_action:
        a = torch.tanh(z)
        lp = lp - torch.log(1 - a * a + 1e-6)
    else:
        a = z
    return a, lp.sum(-1, keepdim=True), mean, log_std


class TwinSACQOracle:
    def __init__(self, pf, q1, q2, plr=3e-4, qlr=3e-4, discount=0.99, tau=0.005, target_entropy=-6.0,
                 w_std=1e-3, w_mean=1e-3, grad_clip=None, act="relu", tanh_action=True, log_alpha=0.0):
        mk = lambda ps: [p.clone().requires_grad_(True) for p in ps]
        self.pf, self.q1, self.q2 = mk(pf), mk(q1), mk(q2)
        self.tq1 = [p.detach().clone() for p in q1]
        self.tq2 = [p.detach().clone() for p in q2]
        self.log_alpha = torch.tensor([log_alpha], requires_grad=True)
        self.pf_opt, self.q1_opt, self.q2_opt = (AdamState(self.pf, plr, eps=1e-8), AdamState(self.q1, qlr, eps=1e-8),
                                                 AdamState(self.q2, qlr, eps=1e-8))
        self.a_opt = AdamState([self.log_alpha], plr, eps=1e-8)
        self.discount, self.tau, self.target_entropy = discount, tau, target_entropy
        self.w_std, self.w_mean, self.grad_clip = w_std, w_mean, grad_clip
        self.act, self.tanh_action = act, tanh_action

    def q(self, params, obs, act):
        return nets.mlp(torch.cat([obs, act], -1), params, self.act)

    def update(self, batch, eps1, eps2):
        f32 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32)
        obs, acts, nobs = f32(batch["obs"]), f32(batch["acts"]), f32(batch["next_obs"])
        rew, term = f32(batch["rewards"]), f32(batch["terminals"])
        new_a, logp, mean, log_std = rsample(nets.mlp(obs, self.pf, self.act), f32(eps1), self.tanh_action)
        q1p, q2p = self.q(self.q1, obs, acts), self.q(self.q2, obs, acts)
        alpha_loss = -(self.log_alpha * (logp + self.target_entropy).detach()).mean()
        g = torch.autograd.grad(alpha_loss, [self.log_alpha])
        self.a_opt.step([self.log_alpha], g)
        alpha = self.log_alpha.exp().detach()
        with torch.no_grad():
            ta, tlogp, _, _ = rsample(nets.mlp(nobs, self.pf, self.act), f32(eps2), self.tanh_action)
            tv = torch.min(self.q(self.tq1, nobs, ta), self.q(self.tq2, nobs, ta)) - alpha * tlogp
        qt = (rew + (1.0 - term) * self.discount * tv).detach()
        l1, l2 = ((q1p - qt) ** 2).mean(), ((q2p - qt) ** 2).mean()
        qn = torch.min(self.q(self.q1, obs, new_a), self.q(self.q2, obs, new_a))
        pl = (alpha * logp - qn).mean() + self.w_std * (log_std ** 2).mean() + self.w_mean * (mean ** 2).mean()
        info = {}
        for name, loss, params, opt in (("pf", pl, self.pf, self.pf_opt), ("qf1", l1, self.q1, self.q1_opt),
                                        ("qf2", l2, self.q2, self.q2_opt)):
            gr = torch.autograd.grad(loss, params, retain_graph=True)
            if self.grad_clip:
                gr, gn = clip_global_norm(gr, self.grad_clip)
                info["Training/%s_grad_norm" % name] = gn
            opt.step(params, gr)
        with torch.no_grad():
            for src, tgt in ((self.q1, self.tq1), (self.q2, self.tq2)):
                for s, t in zip(src, tgt):
                    t.copy_(t * (1.0 - self.tau) + s * self.tau)
        info.update({"Reward_Mean": rew.mean().item(), "Alpha": alpha.item(), "Alpha_loss": alpha_loss.item(),
                     "Training/policy_loss": pl.item(), "Training/qf1_loss": l1.item(), "Training/qf2_loss": l2.item()})
        for key, t in (("log_std", log_std), ("log_probs", logp), ("mean", mean)):
            info.update({key + "/mean": t.mean().item(), key + "/std": t.std().item(),
                         key + "/max": t.max().item(), key + "/min": t.min().item()})
        return info
