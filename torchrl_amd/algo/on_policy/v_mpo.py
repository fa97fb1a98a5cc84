"""V-MPO on the HIP path (reference: torchrl/algo/on_policy/v_mpo.py:11-194).

`update(batch)` keeps the reference's order -- advantage statistics and normalisation, critic step, actor step on
the top half of the minibatch by advantage -- as a fixed launch sequence with an explicit chain rule:
  advantage statistics / normalisation .. trl_adv_stats_f64, trl_adv_normalize_f32
  value net, MSE and its gradient ......... dense-layer kernels (k_gemm.hip) + trl_mse_value_loss_f32
  top half by advantage ................... torch.sort on the device (the one piece of torch arithmetic here) + row gather
  policy and target policy forward ........ dense-layer kernels
  phi = softmax(adv / eta), log pi, KL(pi || pi_target), L_pi and its gradients, the dual variables' gradients,
  their Adam step and clamp ............... trl_vmpo_losses_f32 (eta, alpha and their moments stay on the device)
  policy backward, clip 0.5 + Adam x2 ..... dense-layer kernels + trl_clip_adam_f32 on the flat [pf | vf] buffer
One read-back per update.  Works for any MLP shape (the arbitrary-shape engine of ppo.py)."""
import copy

import numpy as np
import torch

from ... import _C
from .a2c import A2C
from .ppo import _GenericPPO


class VMPO(A2C):
    def __init__(self, pf, opt_epochs=10, eta_eps=0.02, alpha_eps=0.1, clipped_value_loss=False, **kwargs):
        self.target_pf = copy.deepcopy(pf)
        super().__init__(pf=pf, **kwargs)
        self.eta_eps, self.alpha_eps = eta_eps, alpha_eps
        self.opt_epochs = opt_epochs
        self.sample_key = ["obs", "acts", "advs", "estimate_returns", "values"]
        self._engine = None

    @property
    def networks(self):
        return [self.pf, self.vf, self.target_pf]

    @property
    def eta(self):
        return self.engine().dual[0:1]

    @property
    def alpha(self):
        return self.engine().dual[1:2]

    def engine(self):
        if self._engine is None:
            self._engine = _VMPOEngine(self)
        return self._engine

    def update_per_epoch(self):
        self.process_epoch_samples()
        self.engine().sync_target_pf()                                 # target_pf <- pf (v_mpo.py:50)
        for _ in range(self.opt_epochs):
            for batch in self.replay_buffer.one_iteration(self.batch_size, self.sample_key, self.shuffle):
                self.logger.add_update_info(self.update(batch))

    def update(self, batch):
        self.training_update_num += 1
        return self.engine().update(batch)


class _VMPOEngine(_GenericPPO):
    def __init__(self, algo):
        super().__init__(algo)
        if algo.optimizer_class is not torch.optim.Adam:
            raise _C.TrlError("the V-MPO step implements torch.optim.Adam only")
        self.tlayers = self.ops.linear_layers(algo.target_pf)
        # eta, alpha, their Adam moments and step count (v_mpo.py:26-38: eta = 1, alpha = 0.1, Adam(lr = plr, eps 1e-5))
        self.dual = torch.tensor([1.0, 0.1, 0.0, 0.0, 0.0, 0.0, 0.0], device=self.dev)
        self._raw = torch.zeros(4 + 12 + 1 + 1, dtype=torch.float64, device=self.dev)   # adv raw | info | vf loss | norms (2 f32)
        self._zero_idx = torch.zeros(1, 1, dtype=torch.int64, device=self.dev)

    def update(self, batch):
        from ... import dist
        algo, ops, dev = self.algo, self.ops, self.dev
        as_t = lambda x: (x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))) \
            .to(device=dev, dtype=torch.float32).contiguous()
        obs, acts = as_t(batch['obs']), as_t(batch['acts'])
        advs, rets = as_t(batch['advs']).reshape(-1), as_t(batch['estimate_returns']).reshape(-1)
        if dist.collectives_active():
            # The actor step runs on the top half of the GLOBAL minibatch by advantage, which no rank can pick from its
            # env shard alone: gather the shards (4 small tensors) and run the whole update replicated -- every rank
            # then holds the single-process minibatch, so no gradient exchange is needed and parameters stay identical.
            n = int(algo.replay_buffer.env_nums)
            if obs.shape[0] % n != 0:
                raise _C.TrlError("minibatch of %d samples is not a whole number of %d-env rows" % (obs.shape[0], n))
            obs, acts = dist.gather_env_shards(obs, n), dist.gather_env_shards(acts, n)
            advs = dist.gather_env_shards(advs.view(-1, 1), n).view(-1)
            rets = dist.gather_env_shards(rets.view(-1, 1), n).view(-1)
        B = int(obs.shape[0])
        raw, info, vloss = self._raw[:4], self._raw[4:16], self._raw[16:17]
        norms = self._raw[17:18].view(torch.float32)
        ws = self._ws(B)
        _C.adv_stats(advs.view(1, B), self._zero_idx, raw.view(1, 4))
        adv_n = _C.adv_normalize(advs, raw, B)
        # ---- critic (v_mpo.py:136-153) ----
        v, tape_vf = ops.mlp_forward(self.vf_layers, obs, self.act)
        d_v = _C.mse_value_loss(v.view(-1), rets, B, vloss)
        ops.mlp_backward(tape_vf, d_v, grads=self.gviews[1], workspace=ws)
        # ---- actor on the top half by advantage (v_mpo.py:64-70) ----
        idx = torch.sort(adv_n, descending=True).indices.chunk(2, dim=0)[0].contiguous()
        obs_s, acts_s, adv_s = _C.gather_rows(obs, idx), _C.gather_rows(acts, idx), adv_n[idx].contiguous()
        mean, tape_pf = ops.mlp_forward(self.pf_layers, obs_s, self.act)
        tmean, _ = ops.mlp_forward(self.tlayers, obs_s, self.act)
        d_mean = _C.vmpo_losses(mean, tmean, algo.pf.logstd.detach(), algo.target_pf.logstd.detach(), acts_s, adv_s,
                                self.dual, bool(algo.pf.tanh_action), algo.eta_eps, algo.alpha_eps, algo.plr,
                                self.g_logstd, info)
        ops.mlp_backward(tape_pf, d_mean, grads=self.gviews[0], workspace=ws)
        # ---- clip_grad_norm_(0.5) + Adam(eps 1e-5) for both nets ----
        a = _C.AdamArgs()
        a.params, a.grads, a.exp_avg, a.exp_avg_sq = (self.flat.data_ptr(), self.grads.data_ptr(),
                                                      self.m.data_ptr(), self.v.data_ptr())
        a.n_groups = 2
        a.group_sizes[0], a.group_sizes[1] = self.P_pf, self.P_vf
        a.group_lr[0] = algo.pf_optimizer.param_groups[0]['lr']
        a.group_lr[1] = algo.vf_optimizer.param_groups[0]['lr']
        a.max_norm, a.beta1, a.beta2, a.eps, a.grad_scale = 0.5, 0.9, 0.999, 1e-5, 1.0
        self.step_count += 1
        a.step_count, a.norms_out = self.step_count, norms.data_ptr()
        _C.clip_adam(a, dev)
        for s in self._opt_steps:
            s.fill_(float(self.step_count))
        host = self._raw.cpu()                                         # the only host sync of the update
        r, i = host[:4].numpy(), host[4:16].numpy()
        nrm = host[17:18].view(torch.float32).numpy()
        mean_adv = r[0] / B
        return {'advs/mean': mean_adv, 'advs/std': float(np.sqrt(max((r[1] - r[0] * mean_adv) / (B - 1), 0.0))),
                'advs/max': r[2], 'advs/min': -r[3],
                'Training/vf_loss': float(host[16]) / B, 'grad_norm/vf': float(nrm[1]),
                'Training/policy_loss': i[0], 'Training/alpha_loss': i[9], 'Training/alpha': i[10], 'Training/eta': i[11],
                'logprob/mean': i[1], 'logprob/std': i[2], 'logprob/max': i[3], 'logprob/min': i[4],
                'KL/mean': i[5], 'KL/std': i[6], 'KL/max': i[7], 'KL/min': i[8], 'grad_norm/pf': float(nrm[0])}
