"""V-MPO minibatch update, torch-CPU fp32 oracle.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates torchrl/algo/on_policy/v_mpo.py:57-181 over flat parameter lists (same conventions as oracle/ppo.py):
* advantage statistics are logged, then adv = (adv - mean) / (std_unbiased + 1e-5) (:171-177);
* critic first: MSE, clip_grad_norm_(0.5), Adam(eps=1e-5) (:136-153);
* actor on the TOP HALF of the minibatch by normalised advantage (`sort(descending)`, first `chunk(2)`, :64-70):
  phi = softmax(adv / eta), L_pi = mean(-phi * log pi + alpha * KL(pi || pi_target)),
  L_eta = eta * eps_eta + eta * log(mean(exp(adv / eta))), L_alpha = alpha * eps_alpha - alpha * mean(KL);
  one backward of the sum; clip_grad_norm_(0.5) on the policy only; Adam(eps=1e-5) for the policy and Adam(lr=plr,
  eps=1e-5) for (eta, alpha); both are clamped at 1e-8 afterwards (:72-117).
Pinned to the reference's own outputs (tests/golden/vmpo_update.npz, tests/test_oracle_golden.py)."""
import numpy as np
import torch

from . import nets
from .ppo import AdamState, clip_global_norm


class VMPOOracle:
    def __init__(self, pf_params, logstd, vf_params, plr=3e-4, vlr=3e-4, eta_eps=0.02, alpha_eps=0.1, act="tanh",
                 tanh_action=True):
        self.pf = [p.clone().requires_grad_(True) for p in pf_params]
        self.logstd = logstd.clone().requires_grad_(True)
        self.vf = [p.clone().requires_grad_(True) for p in vf_params]
        self.pf_opt = AdamState(self.pf + [self.logstd], plr)
        self.vf_opt = AdamState(self.vf, vlr)
        self.eta = torch.tensor([1.0], requires_grad=True)
        self.alpha = torch.tensor([0.1], requires_grad=True)
        self.param_opt = AdamState([self.eta, self.alpha], plr)
        self.eta_eps, self.alpha_eps = eta_eps, alpha_eps
        self.act, self.tanh_action = act, tanh_action
        self.sync_target()

    def sync_target(self):
        self.tpf = [p.detach().clone() for p in self.pf]
        self.tlogstd = self.logstd.detach().clone()

    def update(self, batch):
        f32 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32)
        obs, acts = f32(batch["obs"]), f32(batch["acts"])
        advs, rets = f32(batch["advs"]), f32(batch["estimate_returns"])
        info = {"advs/mean": advs.mean().item(), "advs/std": advs.std().item(),
                "advs/max": advs.max().item(), "advs/min": advs.min().item()}
        advs = (advs - advs.mean()) / (advs.std() + 1e-5)
        # critic
        v = nets.mlp(obs, self.vf, self.act)
        vf_loss = ((v - rets) ** 2).mean()
        g, gn = clip_global_norm(torch.autograd.grad(vf_loss, self.vf), 0.5)
        self.vf_opt.step(self.vf, g)
        info["Training/vf_loss"], info["grad_norm/vf"] = vf_loss.item(), gn
        # actor: top half by advantage
        _, idx = torch.sort(advs, dim=0, descending=True)
        idx = idx.reshape(-1).long().chunk(2, dim=0)[0]
        obs, acts, advs = obs[idx], acts[idx], advs[idx]
        out = nets.policy_update_terms(obs, acts, self.pf, self.logstd, self.act, self.tanh_action)
        with torch.no_grad():
            tgt = nets.policy_update_terms(obs, acts, self.tpf, self.tlogstd, self.act, self.tanh_action)
        lp = out["log_prob"]
        phis = torch.softmax(advs / self.eta.detach(), dim=0)
        eta_loss = self.eta * self.eta_eps + self.eta * torch.log(torch.mean(torch.exp(advs / self.eta)))
        kl = torch.distributions.kl.kl_divergence(torch.distributions.Normal(out["mean"], out["std"]),
                                                  torch.distributions.Normal(tgt["mean"], tgt["std"])).sum(-1, keepdim=True)
        alpha_loss = self.alpha * self.alpha_eps - self.alpha * kl.detach().mean()
        policy_loss = (-phis * lp + self.alpha.detach() * kl).mean()
        loss = policy_loss + eta_loss + alpha_loss
        grads = torch.autograd.grad(loss, self.pf + [self.logstd, self.eta, self.alpha])
        g_pf, gn = clip_global_norm(grads[:-2], 0.5)
        self.pf_opt.step(self.pf + [self.logstd], g_pf)
        self.param_opt.step([self.eta, self.alpha], list(grads[-2:]))
        with torch.no_grad():
            self.eta.clamp_(min=1e-8)
            self.alpha.clamp_(min=1e-8)
        k = kl.detach()
        info.update({"Training/policy_loss": policy_loss.item(), "Training/alpha_loss": alpha_loss.item(),
                     "Training/alpha": self.alpha.item(), "Training/eta": self.eta.item(),
                     "logprob/mean": lp.mean().item(), "logprob/std": lp.std().item(),
                     "logprob/max": lp.max().item(), "logprob/min": lp.min().item(),
                     "KL/mean": k.mean().item(), "KL/std": k.std().item(), "KL/max": k.max().item(), "KL/min": k.min().item(),
                     "grad_norm/pf": gn})
        return info
