"""DQN / QR-DQN update and the synthetic frame env, CPU oracle.

Restates torchrl/algo/off_policy/dqn.py:38-74 and qrdqn.py:22-74 (with
torchrl/algo/utils.py:5-13, networks/base.py:59-107, nets.py:34-52) over flat parameter lists:
conv trunk (activation after every conv) -> NCHW flatten -> FC (+ activation) -> linear head;
  DQN:   loss = MSE(Q(s).gather(a), r + gamma (1 - d) max_a' Q'(s'))
  QRDQN: theta = Q(s).view(B, A, Q)[a]; a* = argmax_a mean_i Q'(s')[a]; T = r + gamma (1 - d) Q'(s')[a*];
         loss = mean over (b, i, j) of huber(T_i - theta_j) * |tau_j - 1[T_i - theta_j < 0]|
one Adam step (default eps 1e-8), then Polyak(tau).  Frames are uint8 and scaled x/255 - 0.5
(env/atari_wrapper.py:230-240) before the network.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import philox
from .ppo import AdamState

TAG_FRAME = 0x46524D45


def scale_frames(u8):
    return torch.as_tensor(np.asarray(u8)).float() / 255.0 - 0.5


def cnn(x, params, strides, act="tanh"):
    """params: [convW, convb]*n_conv + [fcW, fcb]*n_fc ; x float (B, C, H, W)."""
    f = {"tanh": torch.tanh, "relu": torch.relu}[act]
    n_conv = len(strides)
    for k in range(n_conv):
        x = f(F.conv2d(x, params[2 * k], params[2 * k + 1], stride=strides[k]))
    x = x.reshape(x.shape[0], -1)
    rest = params[2 * n_conv:]
    for k in range(len(rest) // 2 - 1):
        x = f(F.linear(x, rest[2 * k], rest[2 * k + 1]))
    return F.linear(x, rest[-2], rest[-1])


def huber(x, k=1.0):
    return torch.where(x.abs() < k, 0.5 * x.pow(2), k * (x.abs() - 0.5 * k))


class DQNOracle:
    def __init__(self, params, strides, act="tanh", qlr=2.5e-4, discount=0.99, tau=0.005, quantile_num=1, action_num=6):
        self.q = [p.clone().requires_grad_(True) for p in params]
        self.tq = [p.detach().clone() for p in params]
        self.opt = AdamState(self.q, qlr, eps=1e-8)
        self.strides, self.act = strides, act
        self.discount, self.tau, self.Q, self.A = discount, tau, quantile_num, action_num

    def update(self, batch):
        obs, nobs = scale_frames(batch["obs"]), scale_frames(batch["next_obs"])
        f32 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32)
        rew, term = f32(batch["rewards"]).reshape(-1, 1), f32(batch["terminals"]).reshape(-1, 1)
        acts = torch.as_tensor(np.asarray(batch["acts"])).long().reshape(-1)
        B = obs.shape[0]
        q = cnn(obs, self.q, self.strides, self.act)
        with torch.no_grad():
            qn = cnn(nobs, self.tq, self.strides, self.act)
        if self.Q == 1:
            qsa = q.gather(-1, acts[:, None])
            tgt = rew + self.discount * (1 - term) * qn.max(-1, keepdim=True)[0]
            loss = ((qsa - tgt) ** 2).mean()
        else:
            q3, n3 = q.view(B, self.A, self.Q), qn.view(B, self.A, self.Q)
            qsa = q3[torch.arange(B), acts]
            astar = n3.mean(dim=2).max(dim=1)[1]
            tgt = rew + self.discount * (1 - term) * n3[torch.arange(B), astar]
            coef = torch.tensor((2 * np.arange(self.Q) + 1) / (2.0 * self.Q), dtype=torch.float32).view(1, -1)
            diff = tgt.unsqueeze(-1) - qsa.unsqueeze(1)
            loss = (huber(diff) * (coef - (diff.detach() < 0).float()).abs()).mean()
        g = torch.autograd.grad(loss, self.q)
        self.opt.step(self.q, g)
        with torch.no_grad():
            for s, t in zip(self.q, self.tq):
                t.copy_(t * (1.0 - self.tau) + s * self.tau)
        return {"Reward_Mean": rew.mean().item(), "Training/qf_loss": loss.item(), "q_s_a": qsa.mean().item()}


class SynthFrameVecEnvCPU:
    """numpy twin of torchrl_amd.env.SynthFrameVecEnv (frames keyed (env_seed; t, 0, blk, 'FRME'))."""

    def __init__(self, env_nums, frame_shape=(4, 84, 84), action_num=6, horizon=1000):
        self.env_nums, self.frame_shape, self.action_num, self.horizon = env_nums, frame_shape, action_num, horizon
        self.seed(0)

    def seed(self, seed):
        self.env_seed = np.int64(seed) * self.env_nums + np.arange(self.env_nums, dtype=np.int64)

    def _frame(self, t, idx):
        hw = self.frame_shape[1] * self.frame_shape[2]
        nblk = hw // 16
        n = len(idx)
        ctr = np.zeros((n, nblk, 4), dtype=np.uint32)
        ctr[..., 0] = np.asarray(t, dtype=np.int64).astype(np.uint32)[:, None]
        ctr[..., 2] = np.arange(nblk, dtype=np.uint32)[None, :]
        ctr[..., 3] = TAG_FRAME
        key = np.zeros((n, nblk, 2), dtype=np.uint32)
        key[..., 0] = (self.env_seed[idx] & 0xFFFFFFFF).astype(np.uint32)[:, None]
        key[..., 1] = ((self.env_seed[idx] >> 32) & 0xFFFFFFFF).astype(np.uint32)[:, None]
        words = philox.philox4x32_10(ctr, key)                       # (n, nblk, 4) uint32, little endian bytes
        return np.ascontiguousarray(words).view(np.uint8).reshape(n, hw)

    def reset(self, mask=None):
        c = self.frame_shape[0]
        if mask is None:
            self._obs = np.zeros((self.env_nums,) + self.frame_shape, dtype=np.uint8)
            self.t = np.zeros(self.env_nums, dtype=np.int64)
            mask = np.ones(self.env_nums, dtype=bool)
        idx = np.nonzero(mask)[0]
        for k in range(c):
            self._obs[idx, k] = self._frame(np.full(len(idx), k - (c - 1)), idx).reshape((len(idx),) + self.frame_shape[1:])
        self.t[idx] = 0
        return self._obs

    def step(self, acts):
        acts = np.asarray(acts).reshape(-1)
        self.t += 1
        idx = np.arange(self.env_nums)
        new = self._frame(self.t, idx).reshape((self.env_nums,) + self.frame_shape[1:])
        self._obs = np.concatenate([self._obs[:, 1:], new[:, None]], axis=1)
        rew = (acts == (new.reshape(self.env_nums, -1)[:, 0] % self.action_num)).astype(np.float32)
        done = self.t >= self.horizon
        return self._obs, rew[:, None], done[:, None], {"time_limit": done.copy()}


class EpsGreedyOracle:
    """EpsilonGreedyDQNDiscretePolicy.explore, restating torchrl/policies/discrete_policies.py:43-67: epsilon decays
    linearly with the call count until `decay_frames`, the greedy action is the argmax (of the quantile mean for
    quantile_num > 1), and an env takes the random action where its uniform draw -- float32 like the reference's
    torch.Tensor(np.random.rand(...)) -- is below epsilon.  Draws come from the global numpy stream, rand then randint."""

    def __init__(self, start_epsilon, end_epsilon, decay_frames, action_num, quantile_num=1):
        self.start, self.end, self.decay, self.A, self.Q = start_epsilon, end_epsilon, decay_frames, action_num, quantile_num
        self.count, self.epsilon = 0, start_epsilon

    def explore(self, q_values):
        self.count += 1
        if self.count < self.decay:
            self.epsilon = self.start - (self.start - self.end) * (self.count / self.decay)
        else:
            self.epsilon = self.end
        q = np.asarray(q_values, dtype=np.float32)
        if self.Q > 1:
            q = q.reshape(q.shape[0], self.A, self.Q).mean(axis=-1)
        action = q.argmax(axis=-1)[:, None].astype(np.int64)
        r = np.random.rand(*action.shape).astype(np.float32)
        random_action = np.random.randint(low=0, high=self.A, size=action.shape)
        take = r < np.float32(self.epsilon)
        action[take] = random_action[take]
        return action

