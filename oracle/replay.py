"""Time-major ring buffer + on-policy post-processing, numpy float64 oracle.

Restates torchrl/replay_buffers/base.py:4-54 and
torchrl/replay_buffers/on_policy.py:5-95 of the reference:

* storage per key is ``(rows, N, feat)`` float64, ``rows = max_size // N``
  (base.py:14, 22-28); one ``add`` writes row ``top`` of every key and advances
  the ring (base.py:34-37);
* ``sample_rows`` = uniform row sampling with the *legacy global numpy RNG*
  (``np.random.randint(0, size, B // N)``, base.py:39-51);
* ``epoch_minibatches`` = one ``np.random.permutation(rows)`` per pass, then
  consecutive slices of ``B // N`` time rows x all N envs (on_policy.py:72-91);
* ``gae`` / ``discounted_return`` = reverse recurrences of on_policy.py:16-70.

Index streams are bit-exact by construction (same numpy calls, same order).
"""
import numpy as np


class RingOracle:
    def __init__(self, max_replay_buffer_size, env_nums=1, time_limit_filter=False):
        self.env_nums = env_nums
        self.rows = max_replay_buffer_size // env_nums      # base.py:14
        self.top = 0
        self.size = 0
        self.time_limit_filter = time_limit_filter
        self.data = {}

    # base.py:19-37
    def add(self, sample):
        for key, val in sample.items():
            val = np.asarray(val)
            if key not in self.data:
                self.data[key] = np.zeros((self.rows,) + val.shape, dtype=np.float64)
            self.data[key][self.top, ...] = val
        self.top = (self.top + 1) % self.rows
        self.size = min(self.size + 1, self.rows)

    # base.py:39-51
    def sample_rows(self, batch_size, keys):
        assert batch_size % self.env_nums == 0, "batch size should be dividable by env_nums"
        nrows = batch_size // self.env_nums
        idx = np.random.randint(0, self.size, nrows)
        return idx, {k: self._take(k, idx) for k in keys}

    def _take(self, key, idx):
        block = self.data[key][idx]
        return block.reshape((len(idx) * self.env_nums,) + block.shape[2:])

    # on_policy.py:9-14
    def last_row(self, keys):
        return {k: self.data[k][self.rows - 1] for k in keys}

    # on_policy.py:72-91
    def epoch_minibatches(self, batch_size, keys, shuffle):
        assert batch_size % self.env_nums == 0, "batch size should be dividable by env_nums"
        nrows = batch_size // self.env_nums
        order = np.random.permutation(self.rows) if shuffle else np.arange(self.rows)
        for pos in range(0, self.rows, nrows):
            idx = order[pos:pos + nrows]
            yield idx, {k: self._take(k, idx) for k in keys}

    # on_policy.py:16-44
    def gae(self, last_value, gamma, tau):
        adv, ret = gae(self.data["rewards"], self.data["values"], self.data["terminals"],
                       self.data.get("time_limits"), last_value, gamma, tau,
                       self.time_limit_filter)
        self.data["advs"], self.data["estimate_returns"] = adv, ret

    # on_policy.py:46-70
    def discounted_return(self, last_value, gamma):
        adv, ret = discounted_return(self.data["rewards"], self.data["values"],
                                     self.data["terminals"], self.data.get("time_limits"),
                                     last_value, gamma, self.time_limit_filter)
        self.data["advs"], self.data["estimate_returns"] = adv, ret


def gae(rewards, values, terminals, time_limits, last_value, gamma, tau, tl_filter):
    """A_t = delta_t + (1-d_t) gamma tau A_{t+1}  [ * (1 - tl_t) if filtered ];
    delta_t = r_t + (1-d_t) gamma V_{t+1} - V_t;  ret_t = A_t + V_t.
    All arrays (T, N, 1); last_value (N, 1).  float64 (on_policy.py:16-44)."""
    r = np.asarray(rewards, dtype=np.float64)
    v = np.asarray(values, dtype=np.float64)
    d = np.asarray(terminals, dtype=np.float64)
    T = r.shape[0]
    v_next = np.concatenate([v[1:], np.asarray(last_value, dtype=np.float64)[None]], 0)
    adv = np.zeros_like(r)
    run = np.zeros_like(r[0])
    for t in range(T - 1, -1, -1):
        nd = 1.0 - d[t]
        delta = r[t] + nd * gamma * v_next[t] - v[t]
        run = delta + nd * gamma * tau * run
        if tl_filter:
            run = run * (1.0 - np.asarray(time_limits[t], dtype=np.float64))
        adv[t] = run
    return adv, adv + v


def discounted_return(rewards, values, terminals, time_limits, last_value, gamma, tl_filter):
    """R_t = r_t + (1-d_t) gamma R_{t+1} (1-tl_t) + tl_t V_t (filtered) or
    R_t = r_t + (1-d_t) gamma R_{t+1};  adv = R - V  (on_policy.py:46-70)."""
    r = np.asarray(rewards, dtype=np.float64)
    v = np.asarray(values, dtype=np.float64)
    d = np.asarray(terminals, dtype=np.float64)
    T = r.shape[0]
    ret = np.zeros_like(r)
    run = np.asarray(last_value, dtype=np.float64)
    for t in range(T - 1, -1, -1):
        nd = 1.0 - d[t]
        if tl_filter:
            tl = np.asarray(time_limits[t], dtype=np.float64)
            run = r[t] + nd * gamma * run * (1.0 - tl) + tl * v[t]
        else:
            run = r[t] + nd * gamma * run
        ret[t] = run
    return ret - v, ret
