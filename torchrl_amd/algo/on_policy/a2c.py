"""A2C (torchrl/algo/on_policy/a2c.py:8-112): policy / value nets, their two Adam(eps=1e-5) optimisers and
the un-clipped actor-critic update.  The optimiser objects are kept for their `param_groups` (the linear LR
schedule writes there) and `state_dict`; the step itself is done by the fused engine on flat buffers that
the optimiser state aliases.

`update(batch)` = the PPO launch sequence with `loss_mode = TRL_LOSS_A2C`:
L_pi = -mean(log pi(a|s) * adv_normalised) - c_ent * mean(ent), L_v = MSE(V(s), R)   (a2c.py:61-75),
clip_grad_norm_(0.5) + Adam for each net.  `update_per_epoch` keeps the reference's loop
(on_rl_algo.py:35-40: one pass of `one_iteration` minibatches) but runs it through the minibatch row
indices on the device-resident buffer instead of materialising batches."""

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

from ... import _C
from .on_rl_algo import OnRLAlgo


class A2C(OnRLAlgo):
    def __init__(self, pf, vf, plr=3e-4, vlr=3e-4, optimizer_class=optim.Adam, entropy_coeff=0.001, **kwargs):
        super().__init__(**kwargs)
        self.pf = pf
        self.vf = vf
        self.to(self.device)
        self.plr = plr
        self.vlr = vlr
        self.optimizer_class = optimizer_class
        self.pf_optimizer = optimizer_class(self.pf.parameters(), lr=self.plr, eps=1e-5)
        self.vf_optimizer = optimizer_class(self.vf.parameters(), lr=self.vlr, eps=1e-5)
        self.entropy_coeff = entropy_coeff
        self.vf_criterion = nn.MSELoss()

    loss_mode = _C.LOSS_A2C

    def engine(self):
        if getattr(self, "_engine", None) is None:
            from .ppo import make_engine
            self._engine = make_engine(self)
        return self._engine

    def update_per_epoch(self):
        self.process_epoch_samples()
        buf = self.replay_buffer
        row_idx = buf.epoch_row_indices(self.batch_size, self.shuffle)      # one pass (on_rl_algo.py:37-40)
        tensors = {"obs": buf._obs, "acts": buf._acts, "advs": buf._advs, "rets": buf._estimate_returns,
                   "old_values": None, "old_logp": None}
        self._run_and_log(tensors, row_idx, buf.env_nums)

    def _run_and_log(self, tensors, row_idx, n_envs, **prologue):
        """The epoch's minibatch updates through the engine and their info dicts to the logger.  With a logger that takes
        `add_update_infos_later` and the fused engine the updates are launched, not awaited: the logger resolves the
        statistics at its next row (or the engine at its next run, under the next rollout's shadow) -- between two
        iterations the device never waits for the host (`algo.eager_update_infos = True`: read in place)."""
        eng = self.engine()
        later = getattr(self.logger, "add_update_infos_later", None)
        if later is not None and getattr(eng, "defers", False) and not getattr(self, "eager_update_infos", False):
            pending = eng.run(tensors, row_idx, n_envs, defer=True, **prologue)
            self.training_update_num += len(pending)
            later(pending.resolve)
            return
        infos = eng.run(tensors, row_idx, n_envs, **prologue)
        self.training_update_num += len(infos)
        for info in infos:
            self.logger.add_update_info(info)

    def update(self, batch):
        self.training_update_num += 1
        dev = self.device
        as_t = lambda x: (x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))) \
            .to(device=dev, dtype=torch.float32).contiguous()
        obs, acts = as_t(batch['obs']), as_t(batch['acts'])
        B = obs.shape[0]
        tensors = {"obs": obs.reshape(1, B, -1), "acts": acts.reshape(1, B, -1),
                   "advs": as_t(batch['advs']).reshape(1, B, 1), "rets": as_t(batch['estimate_returns']).reshape(1, B, 1),
                   "old_values": None, "old_logp": None}
        return self.engine().run(tensors, np.zeros((1, 1), dtype=np.int64), B)[0]

    @property
    def snapshot_networks(self):
        return [("pf", self.pf), ("vf", self.vf)]
