"""TRPO policy / value updates, torch-CPU fp32 oracle.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates torchrl/algo/on_policy/trpo.py:28-251 over flat parameter lists:
* advantages normalised with eps 1e-4 (:168); surrogate L = -mean(p / (p.detach() + 1e-8) * adv) - c_ent * mean(ent)
  and its gradient g (:170-186);
* step direction x = CG(F, -g): F v = Hessian of mean KL(pi_theta || pi_theta.detach()) times v (double backward) +
  cg_damping * v; double-precision dot products, stop when r.r < residual_tol (:62-108);
* full step = x / sqrt(0.5 x.F x / max_kl); backtracking line search over 0.5^k on -mean(ratio * adv) with accept ratio
  0.1 against the expected improvement -g.x / lm * 0.5^k (:110-152, 188-199);
* value step: 0.5 * MSE, clip_grad_norm_(0.5), Adam(eps 1e-5) (:228-251).
Pinned to the reference's own outputs (tests/golden/trpo_update.npz, tests/test_oracle_golden.py)."""
import numpy as np
import torch

from . import nets
from .ppo import AdamState, clip_global_norm


class TRPOOracle:
    def __init__(self, pf_params, logstd, vf_params, vlr=3e-4, max_kl=0.01, cg_damping=0.1, cg_iters=10, residual_tol=1e-10,
                 entropy_coeff=0.01, act="tanh", tanh_action=False):
        self.pf = [p.clone().requires_grad_(True) for p in pf_params]
        self.logstd = logstd.clone().requires_grad_(True)
        self.vf = [p.clone().requires_grad_(True) for p in vf_params]
        self.vf_opt = AdamState(self.vf, vlr)
        self.max_kl, self.cg_damping, self.cg_iters, self.residual_tol = max_kl, cg_damping, cg_iters, residual_tol
        self.entropy_coeff, self.act, self.tanh_action = entropy_coeff, act, tanh_action

    # ---- flat views ----
    def _params(self):
        return self.pf + [self.logstd]

    def _flat(self):
        return torch.cat([p.detach().reshape(-1) for p in self._params()])

    def _unflat(self, theta):
        out, off = [], 0
        for p in self._params():
            out.append(theta[off:off + p.numel()].view_as(p)); off += p.numel()
        return out[:-1], out[-1]

    def _terms(self, pf, logstd):
        return nets.policy_update_terms(self.obs, self.acts, pf, logstd, self.act, self.tanh_action)

    def _mean_kl(self):
        out = self._terms(self.pf, self.logstd)
        m_old, s_old = out["mean"], out["std"]
        m_new, s_new = m_old.detach(), s_old.detach()
        return torch.mean(torch.sum(torch.log(s_new) - torch.log(s_old)
                                    + (s_old * s_old + (m_old - m_new) ** 2) / (2.0 * s_new * s_new) - 0.5, 1))

    def _fvp(self, v):
        g = torch.autograd.grad(self._mean_kl(), self._params(), create_graph=True)
        gv = torch.sum(torch.cat([x.reshape(-1) for x in g]) * v)
        h = torch.autograd.grad(gv, self._params())
        return torch.cat([x.contiguous().reshape(-1) for x in h]) + self.cg_damping * v.detach()

    def _cg(self, b):
        p, r, x = b.clone(), b.clone(), torch.zeros_like(b)
        rdotr = r.double().dot(r.double())
        for _ in range(self.cg_iters):
            z = self._fvp(p)
            v = (rdotr / p.double().dot(z.double())).float()
            x += v * p
            r -= v * z
            newrdotr = r.double().dot(r.double())
            p = r + (newrdotr / rdotr).float() * p
            rdotr = newrdotr
            if rdotr < self.residual_tol:
                break
        return x

    def _surrogate(self, theta):
        pf, ls = self._unflat(theta.detach())
        with torch.no_grad():
            new = self._terms(pf, ls)["log_prob"]
            old = self._terms(self.pf, self.logstd)["log_prob"]
            return -torch.mean(torch.exp(new - old) * self.advs)

    def _linesearch(self, x, fullstep, expected_improve_rate):
        fval = self._surrogate(x)
        for stepfrac in .5 ** np.arange(10):
            xnew = x + float(stepfrac) * fullstep
            actual = fval - self._surrogate(xnew)
            if actual / (expected_improve_rate * float(stepfrac)) > .1 and actual > 0:
                return xnew
        return x.detach()

    def update(self, batch):
        f32 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32)
        self.obs, self.acts, advs = f32(batch["obs"]), f32(batch["acts"]), f32(batch["advs"])
        info = {"advs/mean": advs.mean().item(), "advs/std": advs.std().item(),
                "advs/max": advs.max().item(), "advs/min": advs.min().item()}
        self.advs = (advs - advs.mean()) / (advs.std() + 1e-4)
        out = self._terms(self.pf, self.logstd)
        lp, ent = out["log_prob"], out["ent"]
        p_new = torch.exp(lp)
        ratio = p_new / (p_new.detach() + 1e-8)
        loss = -torch.mean(ratio * self.advs) - self.entropy_coeff * ent.mean()
        g = torch.cat([x.reshape(-1) for x in torch.autograd.grad(loss, self._params())]).detach()
        if g.nonzero().size()[0]:
            step = self._cg(-g)
            shs = .5 * step.dot(self._fvp(step))
            lm = torch.sqrt(shs / self.max_kl)
            theta = self._linesearch(self._flat(), step / lm, -g.dot(step) / lm)
            if not torch.isnan(theta).any():
                pf, ls = self._unflat(theta)
                with torch.no_grad():
                    for p, q in zip(self._params(), pf + [ls]):
                        p.copy_(q)
        info.update({"Training/policy_loss": loss.item(), "logprob/mean": lp.mean().item(), "logprob/std": lp.std().item(),
                     "logprob/max": lp.max().item(), "logprob/min": lp.min().item()})
        return info

    def update_vf(self, batch):
        f32 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32)
        obs, rets = f32(batch["obs"]), f32(batch["estimate_returns"])
        vf_loss = 0.5 * ((nets.mlp(obs, self.vf, self.act) - rets) ** 2).mean()
        g, gn = clip_global_norm(torch.autograd.grad(vf_loss, self.vf), 0.5)
        self.vf_opt.step(self.vf, g)
        return {"Training/vf_loss": vf_loss.item(), "grad_norm/vf": gn}
