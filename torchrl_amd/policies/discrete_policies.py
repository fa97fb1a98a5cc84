"""Epsilon-greedy wrappers over a Q network (torchrl/policies/discrete_policies.py:23-89).

`explore` keeps the reference's host-side protocol -- linear epsilon decay per call, then
`np.random.rand(*shape)` and `np.random.randint(0, A, shape)` from the global numpy stream (so
seeded runs replay the reference's exploration decisions) -- but the Q network runs on the conv /
dense HIP kernels and the argmax + mask is one kernel (trl_eps_greedy_i64).  Unlike the
reference's QR-DQN policy (single env, `.item()`, its Q16) both policies are vectorised over N
envs; actions are returned as (N, 1) int64.
"""
import numpy as np
import torch

from .. import _C, ops


class EpsilonGreedyDQNDiscretePolicy:
    quantile_num = 1

    def __init__(self, qf, start_epsilon, end_epsilon, decay_frames, action_shape):
        self.qf = qf
        self.start_epsilon = start_epsilon
        self.end_epsilon = end_epsilon
        self.decay_frames = decay_frames
        self.count = 0
        self.action_shape = action_shape
        self.epsilon = self.start_epsilon
        self.continuous = False

    def _q(self, x):
        with torch.no_grad():
            if x.dtype == torch.uint8:
                return ops.cnn_forward(self.qf, x)[0]
            return self.qf(x)

    def q_to_a(self, q):
        return _C.eps_greedy(q.contiguous(), self.action_shape, self.quantile_num, None, None, 0.0).unsqueeze(-1)

    def one_launch_act(self):
        """Conv nets with a hidden FC layer in front of an A <= 8 wide head (Q = 1): head and action are ONE launch."""
        A = int(self.action_shape)
        if self.quantile_num != 1 or A > 8 or not hasattr(self.qf, "base") or not hasattr(self.qf.base, "seq_convs"):
            return False
        fcs = ops.fc_layers(self.qf)
        return len(fcs) >= 2 and int(fcs[-1][0].shape[0]) == A and \
            bool(_C.lib().trl_dqn_act_supported(int(fcs[-1][0].shape[1]), A)) and fcs[-1][0].data_ptr() % 16 == 0

    def act_on(self, x, u=None, rand_act=None, epsilon=0.0, want_q=True, ring_row=None, n_rows=0):
        """(q or None, action (N,) int64) for a batch of observations: greedy (u None) or mixed with the given draws.  Conv
        nets with a hidden FC layer and an A <= 8 wide head (Q = 1) run the head and the action as ONE launch on the last
        hidden activations (trl_dqn_act_f32); everything else is Q network -> trl_eps_greedy_i64.  `ring_row`: the collector's
        device-resident replay row advances in the action launch."""
        A = int(self.action_shape)
        if x.dtype == torch.uint8 and self.one_launch_act():
            with torch.no_grad():
                h, _ = ops.cnn_forward(self.qf, x, head=False)
            w, b = ops.fc_layers(self.qf)[-1]
            if _C.dqn_act_ok(h, w):
                return _C.dqn_act(h, w, b, u, rand_act, epsilon, want_q=want_q, ring_row=ring_row, n_rows=n_rows)
        q = self._q(x)
        return q, _C.eps_greedy(q.contiguous(), A, self.quantile_num, u, rand_act, epsilon, ring_row=ring_row, n_rows=n_rows)

    def explore(self, x):
        self.count += 1
        if x.dim() in (3, 5) and x.shape[0] == 1:                    # the collector's unsqueeze(0) (base.py:185-186)
            x = x.squeeze(0)
        if self.count < self.decay_frames:
            self.epsilon = self.start_epsilon - (self.start_epsilon - self.end_epsilon) * (self.count / self.decay_frames)
        else:
            self.epsilon = self.end_epsilon
        n = int(x.shape[0])
        from .. import dist
        w, r = dist.world_size(), dist.rank()                       # env shards on several ranks: this rank's rows of the
        u = np.random.rand(n * w, 1)[r * n:(r + 1) * n]             # host draws for ALL envs (identical numpy streams)
        ra = np.random.randint(low=0, high=self.action_shape, size=(n * w, 1))[r * n:(r + 1) * n]
        u = torch.from_numpy(u.astype(np.float32)).to(x.device)
        ra = torch.from_numpy(ra.astype(np.int64)).to(x.device)
        output, action = self.act_on(x, u.reshape(-1).contiguous(), ra.reshape(-1).contiguous(), self.epsilon)
        return {"q_value": output, "action": action.unsqueeze(-1)}

    def eval_act(self, x):
        return self.act_on(x, want_q=False)[1].unsqueeze(-1).cpu().numpy()

    def to(self, device):
        self.qf.to(device)

    def parameters(self):
        return self.qf.parameters()


class EpsilonGreedyQRDQNDiscretePolicy(EpsilonGreedyDQNDiscretePolicy):
    """argmax over the mean of the quantiles (discrete_policies.py:86-89), for all N envs."""

    def __init__(self, quantile_num, **kwargs):
        super().__init__(**kwargs)
        self.quantile_num = quantile_num


class CategoricalDisPolicy:
    """Imported by the reference's discrete on-policy examples (ppo / a2c _discrete_atari_vec.py).  There is nothing to
    be faithful to on that path: the reference's PPO.update_actor reads out['log_std'] (ppo.py:52), which this policy's
    `update` does not return (discrete_policies.py:156-168), and the vector collector stores (N, 1) actions that
    `Categorical.log_prob` broadcasts against the (B,) batch shape -- the scripts fail at the first update.  No kernel
    path is built for it; constructing one fails loudly."""

    def __init__(self, *args, **kwargs):
        raise _C.TrlError("CategoricalDisPolicy is not built in torchrl_amd: on-policy algorithms here take "
                          "GuassianContPolicyBasicBias (continuous actions); discrete actions are covered by DQN / QRDQN")
