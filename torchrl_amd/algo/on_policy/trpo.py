"""TRPO on the HIP path (reference: torchrl/algo/on_policy/trpo.py:13-286).

`update(whole_batch)` keeps the reference's procedure -- surrogate gradient, conjugate gradient on the Hessian of the
mean KL, step scaling by max_kl, backtracking line search -- with every pass over the data on the kernels:
  policy forward / backward ............ dense-layer kernels (k_gemm.hip), the tape of the forward is kept
  surrogate loss, its gradients, stats .. trl_trpo_surrogate_f32
  Fisher-vector product F v ............. for a diagonal Gaussian policy the Hessian of mean KL(pi_theta || pi_theta.detach())
      is J^T diag(1 / sigma^2) J / n on the network parameters and 2 on each logstd (trpo.py:62-87 gets the same by
      double backward): one forward-mode pass (trl_linear_fwd_f32 on the v-weights and on the propagated tangents,
      trl_jvp_gate_f32), trl_fisher_scale_f32, one backward pass (trl_linear_bwd_*_f32); + cg_damping * v
  line search objective ................. policy forward with the candidate parameters, trl_gauss_logp_f32, trl_ratio_loss_f32
The CG recurrences themselves are BLAS-1 on ~5 k-element parameter vectors and stay torch tensor expressions on the
device (double-precision dot products as in the reference, :93-106); one scalar read-back per CG iteration and per
line-search trial, as the reference's Python control flow needs.
`update_vf` = 0.5 * MSE, clip_grad_norm_(0.5), Adam(eps 1e-5) on the value net (:228-251).
Works for any MLP shape; policies are used as built (examples/trpo_continuous_vec.py passes no tanh squashing)."""
import numpy as np
import torch

from ... import _C, dist
from .. import utils as atu
from .a2c import A2C
from .ppo import _GenericPPO


class TRPO(A2C):
    def __init__(self, max_kl, cg_damping, v_opt_times, cg_iters, residual_tol, **kwargs):
        super().__init__(**kwargs)
        self.max_kl, self.cg_damping, self.cg_iters, self.residual_tol = max_kl, cg_damping, cg_iters, residual_tol
        self.v_opt_times = v_opt_times
        self.vf_sample_key = ["obs", "estimate_returns"]
        self._engine = None

    @property
    def networks(self):
        return [self.pf, self.vf]

    def engine(self):
        if self._engine is None:
            self._engine = _TRPOEngine(self)
        return self._engine

    def update(self, batch):
        self.training_update_num += 1
        return self.engine().update(batch)

    def update_vf(self, batch):
        self.training_update_num += 1
        return self.engine().update_vf(batch)

    def update_per_epoch(self):
        self.process_epoch_samples()
        atu.update_linear_schedule(self.pf_optimizer, self.current_epoch, self.num_epochs, self.plr)
        atu.update_linear_schedule(self.vf_optimizer, self.current_epoch, self.num_epochs, self.vlr)
        buf = self.replay_buffer
        flat = lambda t: t.reshape(t.shape[0] * t.shape[1], -1)
        self.logger.add_update_info(self.update({"obs": flat(buf._obs), "acts": flat(buf._acts), "advs": flat(buf._advs),
                                                 "estimate_returns": flat(buf._estimate_returns)}))
        for _ in range(self.v_opt_times):
            for batch in buf.one_iteration(self.batch_size, self.vf_sample_key, self.shuffle):
                self.logger.add_update_info(self.update_vf(batch))


class _TRPOEngine(_GenericPPO):
    def __init__(self, algo):
        super().__init__(algo)
        self._stat = torch.zeros(4 + 5 + 1 + 1, dtype=torch.float64, device=self.dev)   # adv raw | info | scalar | vf loss
        self._zero_idx = torch.zeros(1, 1, dtype=torch.int64, device=self.dev)
        self._vf_norm = torch.zeros(1, device=self.dev)
        self.vf_steps = 0

    # ---- flat parameter vectors <-> per-layer views ----
    def _views(self, flat):
        """[(W, b), ...] and the logstd slice of a flat policy-parameter-shaped vector."""
        out, off = [], 0
        for w, b in self.pf_layers:
            out.append((flat[off:off + w.numel()].view(w.shape), flat[off + w.numel():off + w.numel() + b.numel()]))
            off += w.numel() + b.numel()
        return out, flat[off:off + self.A]

    def _forward(self, layers):
        return self.ops.mlp_forward(layers, self.obs, self.act)

    def _logp(self, layers, logstd):
        mean, _ = self._forward(layers)
        return _C.gauss_logp(mean, self.acts, logstd.contiguous(), self.tanh_action)

    def _fvp(self, v):
        """F v + cg_damping * v at the current parameters (tape of the current forward in self.tape)."""
        tape, act = self.tape, self.act
        vl, v_ls = self._views(v)
        n_layers = len(self.pf_layers)
        dh = None
        for k, (w, _b) in enumerate(self.pf_layers):
            inp = tape.x if k == 0 else tape.outs[k - 1]
            a = _C.linear_fwd(inp, vl[k][0].contiguous(), vl[k][1].contiguous(), _C.ACT_NONE)   # x W_v^T + b_v
            b = None if dh is None else _C.linear_fwd(dh, w, None, _C.ACT_NONE)                # dx W^T
            last = k == n_layers - 1
            dh = _C.jvp_gate(a, b, None if last else tape.outs[k], act)
        g_mu = _C.fisher_scale(dh, self.algo.pf.logstd.detach())                                # d KL / d mean
        out = torch.zeros_like(v)
        ol, o_ls = self._views(out)
        self.ops.mlp_backward(tape, g_mu, grads=ol, workspace=self._ws(self.n))
        raw = self.algo.pf.logstd.detach()
        o_ls.copy_(2.0 * v_ls * ((raw >= -20.0) & (raw <= 2.0)).to(v.dtype))
        return out + self.algo.cg_damping * v

    def _cg(self, b):
        algo = self.algo
        p, r, x = b.clone(), b.clone(), torch.zeros_like(b)
        rdotr = r.double().dot(r.double())
        for _ in range(algo.cg_iters):
            z = self._fvp(p)
            v = (rdotr / p.double().dot(z.double())).float()
            x += v * p
            r -= v * z
            newrdotr = r.double().dot(r.double())
            p = r + (newrdotr / rdotr).float() * p
            rdotr = newrdotr
            if float(rdotr) < algo.residual_tol:
                break
        return x

    def _surrogate(self, theta):
        layers, ls = self._views(theta)
        lp_new = self._logp([(w.contiguous(), b.contiguous()) for w, b in layers], ls)
        return float(_C.ratio_loss(lp_new, self.lp_old, self.adv_n, self._stat[9:10]).item())

    def _linesearch(self, x, fullstep, expected_improve_rate):
        fval = self._surrogate(x)
        for stepfrac in .5 ** np.arange(10):
            xnew = x + float(stepfrac) * fullstep
            actual = fval - self._surrogate(xnew)
            if actual / (expected_improve_rate * float(stepfrac)) > .1 and actual > 0:
                return xnew
        return x

    def update(self, batch):
        algo, dev = self.algo, self.dev
        as_t = lambda x: (x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))) \
            .to(device=dev, dtype=torch.float32).contiguous()
        self.obs, self.acts = as_t(batch['obs']), as_t(batch['acts'])
        advs = as_t(batch['advs']).reshape(-1)
        if dist.collectives_active():
            # The natural-gradient step (CG on Fisher products, line search) is one global solve: gather the env shards
            # and run it replicated on the single-process batch -- identical parameters on every rank, no exchange.
            ne = int(algo.replay_buffer.env_nums)
            self.obs, self.acts = dist.gather_env_shards(self.obs, ne), dist.gather_env_shards(self.acts, ne)
            advs = dist.gather_env_shards(advs.view(-1, 1), ne).view(-1)
        self.n = n = int(self.obs.shape[0])
        self.tanh_action = bool(getattr(algo.pf, "tanh_action", False))
        raw, info = self._stat[:4], self._stat[4:9]
        _C.adv_stats(advs.view(1, n), self._zero_idx, raw.view(1, 4))
        self.adv_n = _C.adv_normalize(advs, raw, n, eps=1e-4)            # trpo.py:168
        logstd = algo.pf.logstd.detach()
        mean, self.tape = self._forward(self.pf_layers)
        g = torch.zeros(self.P_pf, device=dev)
        gl, g_ls = self._views(g)
        d_mean = _C.trpo_surrogate(mean, logstd, self.acts, self.adv_n, self.tanh_action, algo.entropy_coeff, g_ls, info)
        self.ops.mlp_backward(self.tape, d_mean, grads=gl, workspace=self._ws(n))
        self.lp_old = _C.gauss_logp(mean, self.acts, logstd, self.tanh_action)
        if bool((g != 0).any()):                                          # "ensure gradient is not zero" (:188)
            step = self._cg(-g)
            shs = .5 * step.dot(self._fvp(step))
            lm = torch.sqrt(shs / algo.max_kl)
            theta0 = self.flat[:self.P_pf].clone()
            theta = self._linesearch(theta0, step / lm, float(-g.dot(step) / lm))
            if not bool(torch.isnan(theta).any()):
                self.flat[:self.P_pf].copy_(theta)
        host = self._stat.cpu()
        r, i = host[:4].numpy(), host[4:9].numpy()
        mean_adv = r[0] / n
        return {'advs/mean': mean_adv, 'advs/std': float(np.sqrt(max((r[1] - r[0] * mean_adv) / (n - 1), 0.0))),
                'advs/max': r[2], 'advs/min': -r[3], 'Training/policy_loss': i[0],
                'logprob/mean': i[1], 'logprob/std': i[2], 'logprob/max': i[3], 'logprob/min': i[4]}

    def update_vf(self, batch):
        algo, dev, ops = self.algo, self.dev, self.ops
        as_t = lambda x: (x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))) \
            .to(device=dev, dtype=torch.float32).contiguous()
        obs, rets = as_t(batch['obs']), as_t(batch['estimate_returns']).reshape(-1)
        if dist.collectives_active():
            ne = int(algo.replay_buffer.env_nums)
            obs, rets = dist.gather_env_shards(obs, ne), dist.gather_env_shards(rets.view(-1, 1), ne).view(-1)
        B = int(obs.shape[0])
        v, tape = ops.mlp_forward(self.vf_layers, obs, self.act)
        d_v = _C.mse_value_loss(v.view(-1), rets, 2 * B, self._stat[10:11])       # 0.5 * mean((v - R)^2): d = (v - R) / B
        ops.mlp_backward(tape, d_v, grads=self.gviews[1], workspace=self._ws(B))
        self.vf_steps += 1
        a = _C.AdamArgs()
        o = 4 * self.P_pf
        a.params, a.grads = self.flat.data_ptr() + o, self.grads.data_ptr() + o
        a.exp_avg, a.exp_avg_sq = self.m.data_ptr() + o, self.v.data_ptr() + o
        a.n_groups = 1
        a.group_sizes[0] = self.P_vf
        a.group_lr[0] = algo.vf_optimizer.param_groups[0]['lr']
        a.max_norm, a.beta1, a.beta2, a.eps, a.grad_scale = 0.5, 0.9, 0.999, 1e-5, 1.0
        a.step_count, a.norms_out = self.vf_steps, self._vf_norm.data_ptr()
        _C.clip_adam(a, dev)
        return {'Training/vf_loss': 0.5 * float(self._stat[10].item()) / B, 'grad_norm/vf': float(self._vf_norm.item())}
