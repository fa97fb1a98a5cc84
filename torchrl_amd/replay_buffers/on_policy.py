"""On-policy post-processing on the GPU (torchrl/replay_buffers/on_policy.py:5-95).

`generalized_advantage_estimation` / `discount_reward` run the LDS-staged wave
scan (trl_gae_f32 / trl_discount_reward_f32) over the whole ring -- like the
reference they assume the buffer is exactly full and epoch aligned (its Q3).
`one_iteration` consumes ONE `np.random.permutation(rows)` per call from the
global numpy stream (on_policy.py:76-78) and yields dicts of `B // N` time rows
x all N envs gathered on the GPU.  `epoch_row_indices` exposes the same index
stream without gathering, for the fused PPO path.
"""
import numpy as np
import torch

from .. import _C
from .base import BaseReplayBuffer


class OnPolicyReplayBufferBase:
    def last_sample(self, sample_key):
        return {key: getattr(self, "_" + key)[self._max_replay_buffer_size - 1] for key in sample_key}

    def _scan_inputs(self):
        rows, n = self._max_replay_buffer_size, self.env_nums
        flat = lambda k: getattr(self, "_" + k).reshape(rows, n)
        tl = flat("time_limits") if self.time_limit_filter else None
        self._ensure_key("advs", (n, 1))
        self._ensure_key("estimate_returns", (n, 1))
        return flat("rewards"), flat("values"), flat("terminals"), tl, flat("advs"), flat("estimate_returns")

    @staticmethod
    def _vec(x, device):
        t = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))
        return t.to(device=device, dtype=torch.float32).reshape(-1).contiguous()

    def generalized_advantage_estimation(self, last_value, gamma, tau, last_terminal=None):
        r, v, d, tl, adv, ret = self._scan_inputs()
        lt = None if last_terminal is None else self._vec(last_terminal, r.device)
        _C.gae(r, v, d, tl, self._vec(last_value, r.device), adv, ret, gamma, tau, self.time_limit_filter,
               last_terminal=lt)

    def discount_reward(self, last_value, gamma, last_terminal=None):
        r, v, d, tl, adv, ret = self._scan_inputs()
        lt = None if last_terminal is None else self._vec(last_terminal, r.device)
        _C.discount_reward(r, v, d, tl, self._vec(last_value, r.device), adv, ret, gamma, self.time_limit_filter,
                           last_terminal=lt)

    def epoch_row_indices(self, batch_size, shuffle):
        """(n_minibatches, B // N) int64 host array: the rows `one_iteration` would visit."""
        nrows = self._rows_per_batch(batch_size)
        rows = self._max_replay_buffer_size
        order = np.random.permutation(rows) if shuffle else np.arange(rows)
        if rows % nrows != 0:
            raise ValueError("buffer rows (%d) must be a multiple of batch_size // env_nums (%d)" % (rows, nrows))
        return order.reshape(rows // nrows, nrows).astype(np.int64)

    def one_iteration(self, batch_size, sample_key, shuffle):
        nrows = self._rows_per_batch(batch_size)
        rows = self._max_replay_buffer_size
        order = np.random.permutation(rows) if shuffle else np.arange(rows)
        order_dev = torch.from_numpy(order.astype(np.int64)).to(self._device())
        for pos in range(0, rows, nrows):
            idx = order_dev[pos:pos + nrows].contiguous()
            yield {key: self._gather(key, idx) for key in sample_key}


class OnPolicyReplayBuffer(OnPolicyReplayBufferBase, BaseReplayBuffer):
    pass
