"""A2C base: owns policy / value nets and their two Adam(eps=1e-5) optimisers
(torchrl/algo/on_policy/a2c.py:8-43).  The optimiser objects are kept for their
`param_groups` (the linear LR schedule writes there) and `state_dict`; the step
itself is done by trl_clip_adam_f32 on flat buffers that the optimiser state
aliases.  A2C's own un-clipped update is not on this repo's hot path."""
import torch.nn as nn
import torch.optim as optim

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

    def update(self, batch):
        raise NotImplementedError("A2C.update (a2c.py:45-106) is not on the accelerated path of this build; use PPO")

    @property
    def snapshot_networks(self):
        return [("pf", self.pf), ("vf", self.vf)]
