"""DDPG and TD3 updates, torch-CPU fp32 oracle (test infrastructure only).

Restates torchrl/algo/off_policy/ddpg.py:42-110 and td3.py:57-154 over flat parameter lists, with the
deterministic policies of torchrl/policies/continuous_policy.py:28-74 (forward = [tanh](mlp(x));
FixGuassianContPolicy.explore adds N(0, norm_std_explore) drawn from the CPU generator):

  DDPG:  policy loss = -mean(Q(s, pi(s)));  q_target = r + (1 - d) gamma Q'(s', pi'(s'));  qf loss = MSE
         steps: pf, qf (Adam, default eps 1e-8, optional clip_grad_norm_), Polyak of BOTH pf and qf targets.
  TD3:   a' = clamp(pi'.explore(s') + clamp(N(0, sigma_p), +-c), -1, 1)   (two CPU draws: explore noise, then
         smoothing noise);  q_target = r + (1 - d) gamma min(Q1', Q2')(s', a');  two MSE losses and steps;
         the policy step -mean(Q1(s, pi(s))) and the Polyak update of pf, qf1, qf2 happen only when
         training_update_num % policy_update_delay != 0 (td3.py:124 -- the reference's condition as written).
"""
import numpy as np
import torch

from . import nets
from .ppo import AdamState, clip_global_norm


def det_policy(obs, params, act, tanh_action):
    out = nets.mlp(obs, params, act)
    return torch.tanh(out) if tanh_action else out


class _DetACBase:
    def _q(self, params, obs, act):
        return nets.mlp(torch.cat([obs, act], -1), params, self.act)

    def _step(self, name, loss, params, opt, info, retain=True):
        g = torch.autograd.grad(loss, params, retain_graph=retain)
        if self.grad_clip:
            g, gn = clip_global_norm(g, self.grad_clip)
            info["Training/%s_grad_norm" % name] = gn
        opt.step(params, g)

    def _polyak(self, pairs):
        with torch.no_grad():
            for src, tgt in pairs:
                for s, t in zip(src, tgt):
                    t.copy_(t * (1.0 - self.tau) + s * self.tau)


class DDPGOracle(_DetACBase):
    def __init__(self, pf, qf, plr=3e-4, qlr=3e-4, discount=0.99, tau=0.005, grad_clip=None, act="relu", tanh_action=True):
        mk = lambda ps: [p.clone().requires_grad_(True) for p in ps]
        self.pf, self.qf = mk(pf), mk(qf)
        self.tpf, self.tqf = [p.detach().clone() for p in pf], [p.detach().clone() for p in qf]
        self.pf_opt, self.qf_opt = AdamState(self.pf, plr, eps=1e-8), AdamState(self.qf, qlr, eps=1e-8)
        self.discount, self.tau, self.grad_clip, self.act, self.tanh_action = discount, tau, grad_clip, act, tanh_action

    def update(self, batch):
        f32 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32)
        obs, acts, nobs = f32(batch["obs"]), f32(batch["acts"]), f32(batch["next_obs"])
        rew, term = f32(batch["rewards"]), f32(batch["terminals"])
        new_a = det_policy(obs, self.pf, self.act, self.tanh_action)
        pl = -self._q(self.qf, obs, new_a).mean()
        with torch.no_grad():
            tq = self._q(self.tqf, nobs, det_policy(nobs, self.tpf, self.act, self.tanh_action))
        qt = rew + (1.0 - term) * self.discount * tq
        ql = ((self._q(self.qf, obs, acts) - qt) ** 2).mean()
        info = {}
        self._step("pf", pl, self.pf, self.pf_opt, info)
        self._step("qf", ql, self.qf, self.qf_opt, info)
        self._polyak(((self.pf, self.tpf), (self.qf, self.tqf)))
        info.update({"Reward_Mean": rew.mean().item(), "Training/policy_loss": pl.item(), "Training/qf_loss": ql.item(),
                     "new_actions/mean": new_a.mean().item(), "new_actions/std": new_a.std().item(),
                     "new_actions/max": new_a.max().item(), "new_actions/min": new_a.min().item()})
        return info


class TD3Oracle(_DetACBase):
    def __init__(self, pf, qf1, qf2, plr=3e-4, qlr=3e-4, discount=0.99, tau=0.005, grad_clip=None, act="relu",
                 tanh_action=True, policy_update_delay=2, norm_std_policy=0.2, noise_clip=0.5, norm_std_explore=0.1):
        mk = lambda ps: [p.clone().requires_grad_(True) for p in ps]
        self.pf, self.q1, self.q2 = mk(pf), mk(qf1), mk(qf2)
        self.tpf = [p.detach().clone() for p in pf]
        self.tq1, self.tq2 = [p.detach().clone() for p in qf1], [p.detach().clone() for p in qf2]
        self.pf_opt = AdamState(self.pf, plr, eps=1e-8)
        self.q1_opt, self.q2_opt = AdamState(self.q1, qlr, eps=1e-8), AdamState(self.q2, qlr, eps=1e-8)
        self.discount, self.tau, self.grad_clip, self.act, self.tanh_action = discount, tau, grad_clip, act, tanh_action
        self.policy_update_delay, self.norm_std_policy, self.noise_clip = policy_update_delay, norm_std_policy, noise_clip
        self.norm_std_explore = norm_std_explore
        self.training_update_num = 0

    def update(self, batch, eps_explore, eps_smooth):
        """eps_*: the two standard-normal draws of this update (B, A) -- explore noise of target_pf, then smoothing."""
        self.training_update_num += 1
        f32 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32)
        obs, acts, nobs = f32(batch["obs"]), f32(batch["acts"]), f32(batch["next_obs"])
        rew, term = f32(batch["rewards"]), f32(batch["terminals"])
        with torch.no_grad():
            ta = det_policy(nobs, self.tpf, self.act, self.tanh_action) + self.norm_std_explore * f32(eps_explore)
            ta = torch.clamp(ta + torch.clamp(self.norm_std_policy * f32(eps_smooth), -self.noise_clip, self.noise_clip), -1, 1)
            tq = torch.min(self._q(self.tq1, nobs, ta), self._q(self.tq2, nobs, ta))
        qt = rew + (1.0 - term) * self.discount * tq
        l1 = ((self._q(self.q1, obs, acts) - qt) ** 2).mean()
        l2 = ((self._q(self.q2, obs, acts) - qt) ** 2).mean()
        info = {}
        self._step("qf1", l1, self.q1, self.q1_opt, info)
        self._step("qf2", l2, self.q2, self.q2_opt, info)
        info.update({"Reward_Mean": rew.mean().item(), "Training/qf1_loss": l1.item(), "Training/qf2_loss": l2.item()})
        if self.training_update_num % self.policy_update_delay:
            new_a = det_policy(obs, self.pf, self.act, self.tanh_action)
            pl = -self._q(self.q1, obs, new_a).mean()                    # Q1 AFTER its step (td3.py:128-130)
            self._step("pf", pl, self.pf, self.pf_opt, info)
            self._polyak(((self.pf, self.tpf), (self.q1, self.tq1), (self.q2, self.tq2)))
            info.update({"Training/policy_loss": pl.item(),
                         "new_actions/mean": new_a.mean().item(), "new_actions/std": new_a.std().item(),
                         "new_actions/max": new_a.max().item(), "new_actions/min": new_a.min().item()})
        return info
