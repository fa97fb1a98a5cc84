"""PPO minibatch update / epoch driver, torch-CPU fp32 oracle.

Restates torchrl/algo/on_policy/ppo.py:27-152 (+ a2c.py:29-43,
on_rl_algo.py:22-33, algo/utils.py:23-32) as a small state object over flat
parameter lists, with Adam and global-norm clipping written out explicitly:

* advantage stats are logged, then ``adv = (adv - mean) / (std_unbiased + 1e-5)``
  per minibatch (ppo.py:141-147);
* critic step first: MSE (a2c.py:43) or clipped value loss (ppo.py:104-111),
  ``clip_grad_norm_(0.5)``, Adam(eps=1e-5) (ppo.py:93-122);
* actor step: ratio = exp(logp - logp_old) with logp_old from the frozen
  epoch-start copy ``target_pf`` (ppo.py:54-56), clipped surrogate, entropy
  bonus, ``clip_grad_norm_(0.5)``, Adam(eps=1e-5) (ppo.py:41-91);
* ``epoch``: last_value = vf(next_obs[T-1]) * (1 - terminal[T-1])
  (on_rl_algo.py:22-27), GAE, lr = lr0 * (1 - epoch/num_epochs) for both
  optimisers (utils.py:28-32), target_pf <- pf, then ``opt_epochs`` passes of
  ``epoch_minibatches`` (ppo.py:27-39).
"""
import math
import numpy as np
import torch
from . import nets


class AdamState:
    """torch.optim.Adam(betas=(0.9, 0.999), eps, weight_decay=0) written out."""
    def __init__(self, params, lr, eps=1e-5, betas=(0.9, 0.999)):
        self.lr, self.eps, self.b1, self.b2 = lr, eps, betas[0], betas[1]
        self.t = 0
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]

    def step(self, params, grads):
        self.t += 1
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        step_size = self.lr / bc1
        for p, g, m, v in zip(params, grads, self.m, self.v):
            m.mul_(self.b1).add_(g, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.data.addcdiv_(m, denom, value=-step_size)


def clip_global_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1."""
    total = torch.sqrt(sum((g.float() ** 2).sum() for g in grads))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return [g * coef for g in grads], float(total)


class PPOOracle:
    def __init__(self, pf_params, logstd, vf_params, plr=3e-4, vlr=3e-4,
                 entropy_coeff=0.001, clip_para=0.2, opt_epochs=10,
                 clipped_value_loss=False, act="tanh", tanh_action=True,
                 discount=0.99, tau=0.95, num_epochs=3000, batch_size=128,
                 shuffle=True, gae=True):
        self.pf = [p.clone().requires_grad_(True) for p in pf_params]
        self.logstd = logstd.clone().requires_grad_(True)
        self.vf = [p.clone().requires_grad_(True) for p in vf_params]
        self.plr, self.vlr = plr, vlr
        self.pf_opt = AdamState(self.pf + [self.logstd], plr)
        self.vf_opt = AdamState(self.vf, vlr)
        self.entropy_coeff, self.clip_para = entropy_coeff, clip_para
        self.opt_epochs, self.clipped_value_loss = opt_epochs, clipped_value_loss
        self.act, self.tanh_action = act, tanh_action
        self.discount, self.tau = discount, tau
        self.num_epochs, self.batch_size = num_epochs, batch_size
        self.shuffle, self.use_gae = shuffle, gae
        self.sync_target()

    def sync_target(self):
        self.tpf = [p.detach().clone() for p in self.pf]
        self.tlogstd = self.logstd.detach().clone()

    # ppo.py:124-152
    def update(self, batch):
        f32 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32)
        obs, acts = f32(batch["obs"]), f32(batch["acts"])
        advs, old_v, rets = f32(batch["advs"]), f32(batch["values"]), f32(batch["estimate_returns"])
        info = {"advs/mean": advs.mean().item(), "advs/std": advs.std().item(),
                "advs/max": advs.max().item(), "advs/min": advs.min().item()}
        advs = (advs - advs.mean()) / (advs.std() + 1e-5)

        # critic (ppo.py:93-122)
        v = nets.mlp(obs, self.vf, self.act)
        if self.clipped_value_loss:
            v_clip = old_v + (v - old_v).clamp(-self.clip_para, self.clip_para)
            vf_loss = 0.5 * torch.max((v - rets) ** 2, (v_clip - rets) ** 2).mean()
        else:
            vf_loss = ((v - rets) ** 2).mean()
        g = torch.autograd.grad(vf_loss, self.vf)
        g, gn = clip_global_norm(g, 0.5)
        self.vf_opt.step(self.vf, g)
        info["Training/vf_loss"] = vf_loss.item()
        info["grad_norm/vf"] = gn

        # actor (ppo.py:41-91)
        out = nets.policy_update_terms(obs, acts, self.pf, self.logstd, self.act, self.tanh_action)
        with torch.no_grad():
            old = nets.policy_update_terms(obs, acts, self.tpf, self.tlogstd, self.act, self.tanh_action)
        lp = out["log_prob"]
        ratio = torch.exp(lp - old["log_prob"])
        s1 = ratio * advs
        s2 = torch.clamp(ratio, 1.0 - self.clip_para, 1.0 + self.clip_para) * advs
        pl = -torch.min(s2, s1).mean() - self.entropy_coeff * out["ent"].mean()
        g = torch.autograd.grad(pl, self.pf + [self.logstd])
        g, gn = clip_global_norm(g, 0.5)
        self.pf_opt.step(self.pf + [self.logstd], g)
        ls = out["log_std"]
        info.update({
            "Training/policy_loss": pl.item(),
            "logprob/mean": lp.mean().item(), "logprob/std": lp.std().item(),
            "logprob/max": lp.max().item(), "logprob/min": lp.min().item(),
            "log_std/mean": ls.mean().item(), "log_std/std": ls.std().item(),
            "log_std/max": ls.max().item(), "log_std/min": ls.min().item(),
            "ratio/max": ratio.max().item(), "ratio/min": ratio.min().item(),
            "grad_norm/pf": gn})
        return info

    # on_rl_algo.py:22-33
    def process_epoch_samples(self, ring):
        last = ring.last_row(["next_obs", "terminals", "time_limits"])
        with torch.no_grad():
            lv = nets.mlp(torch.as_tensor(last["next_obs"], dtype=torch.float32),
                          self.vf, self.act).numpy()
        lv = lv * (1 - last["terminals"])
        if self.use_gae:
            ring.gae(lv, self.discount, self.tau)
        else:
            ring.discounted_return(lv, self.discount)

    # ppo.py:27-39
    def epoch(self, ring, current_epoch):
        self.process_epoch_samples(ring)
        frac = current_epoch / float(self.num_epochs)
        self.pf_opt.lr = self.plr - self.plr * frac
        self.vf_opt.lr = self.vlr - self.vlr * frac
        self.sync_target()
        infos = []
        keys = ["obs", "acts", "advs", "estimate_returns", "values"]
        for _ in range(self.opt_epochs):
            for _idx, batch in ring.epoch_minibatches(self.batch_size, keys, self.shuffle):
                infos.append(self.update(batch))
        return infos


class A2COracle(PPOOracle):
    """A2C.update (torchrl/algo/on_policy/a2c.py:45-106) and OnRLAlgo.update_per_epoch
    (on_rl_algo.py:35-40): L_pi = -mean(log pi * adv_normalised) - c_ent * mean(ent), policy step, then
    L_v = MSE(V, R), value step; each with clip_grad_norm_(0.5) and Adam(eps=1e-5)."""

    def update(self, batch):
        f32 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32)
        obs, acts = f32(batch["obs"]), f32(batch["acts"])
        advs, rets = f32(batch["advs"]), f32(batch["estimate_returns"])
        out = nets.policy_update_terms(obs, acts, self.pf, self.logstd, self.act, self.tanh_action)
        lp, ent, std = out["log_prob"], out["ent"], out["std"]
        advs = (advs - advs.mean()) / (advs.std() + 1e-5)
        pl = (-lp * advs).mean() - self.entropy_coeff * ent.mean()
        v = nets.mlp(obs, self.vf, self.act)
        vf_loss = ((v - rets) ** 2).mean()
        g = torch.autograd.grad(pl, self.pf + [self.logstd])
        g, _ = clip_global_norm(g, 0.5)
        self.pf_opt.step(self.pf + [self.logstd], g)
        g = torch.autograd.grad(vf_loss, self.vf)
        g, _ = clip_global_norm(g, 0.5)
        self.vf_opt.step(self.vf, g)
        return {"Training/policy_loss": pl.item(), "Training/vf_loss": vf_loss.item(),
                "v_pred/mean": v.mean().item(), "v_pred/std": v.std().item(),
                "v_pred/max": v.max().item(), "v_pred/min": v.min().item(),
                "std/mean": std.mean().item(), "std/std": std.std().item(),
                "std/max": std.max().item(), "std/min": std.min().item(),
                "ent": ent.mean().item(), "log_prob": lp.mean().item()}

    def epoch(self, ring, current_epoch=0):
        self.process_epoch_samples(ring)
        infos = []
        for _idx, batch in ring.epoch_minibatches(self.batch_size, ["obs", "acts", "advs", "estimate_returns"], self.shuffle):
            infos.append(self.update(batch))
        return infos
