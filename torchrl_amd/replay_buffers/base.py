"""Device-resident time-major ring buffer with the reference's interface
(torchrl/replay_buffers/base.py:4-54).

Layout: one tensor per key, ``_<key>[rows, N, feat]`` with
``rows = max_replay_buffer_size // env_nums`` (base.py:14), fp32 on the GPU
(the reference allocates float64 numpy, base.py:26-27; uint8 frames stay uint8).
`_top` / `_size` are host integers exactly as in the reference.  The uniform
sample draws its row indices with the legacy global numpy RNG
(`np.random.randint(0, size, B // N)`, base.py:44) so the index stream is
bit-exact, uploads the few int64s and gathers on the GPU (trl_gather_rows_*).
"""
import numpy as np
import torch

from .. import _C


class BaseReplayBuffer:
    def __init__(self, max_replay_buffer_size, env_nums=1, time_limit_filter=False, device=None):
        self.env_nums = env_nums
        self._max_replay_buffer_size = max_replay_buffer_size // self.env_nums
        self._top = 0
        self._size = 0
        self.time_limit_filter = time_limit_filter
        self.device = torch.device(device) if device is not None else None
        self._keys = []

    # ---- storage ----
    def _device(self):
        if self.device is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        return self.device

    def _ensure_key(self, key, feat_shape, dtype=torch.float32):
        """Allocate `_key[rows, *feat_shape]` on first use (reference: lazily in add_sample)."""
        name = "_" + key
        if not hasattr(self, name):
            setattr(self, name, torch.zeros((self._max_replay_buffer_size,) + tuple(feat_shape),
                                            dtype=dtype, device=self._device()))
            self._keys.append(key)
        return getattr(self, name)

    def _flip_key(self, key):
        """Swap `_key` with its shadow tensor (allocated on the first flip): a writer that is about to rewrite EVERY row of
        the key gets fresh storage while a reader on another stream still walks the old one (the value function's update
        chain reads `obs` beside the next rollout, algo/on_policy/ppo.py).  Only for full-ring rewrites -- the shadow's
        content is whatever it held two rollouts ago."""
        name, alt = "_" + key, "_" + key + "_shadow"
        cur = getattr(self, name)
        other = getattr(self, alt, None)
        if other is None or other.shape != cur.shape or other.device != cur.device:
            other = torch.zeros_like(cur)
        setattr(self, alt, cur)
        setattr(self, name, other)
        return other

    def _as_row(self, value):
        if isinstance(value, torch.Tensor):
            t = value
        else:
            t = torch.as_tensor(np.asarray(value))
        if t.dtype != torch.uint8:
            t = t.to(torch.float32)
        return t

    def add_sample(self, sample_dict, **kwargs):
        for key, value in sample_dict.items():
            row = self._as_row(value)
            store = self._ensure_key(key, row.shape, row.dtype)
            store[self._top].copy_(row, non_blocking=True)
        self._advance()

    def terminate_episode(self):
        pass

    def _advance(self, steps=1):
        self._boot_fresh = False                                         # (a fused rollout sets it again after its advance)
        self._top = (self._top + steps) % self._max_replay_buffer_size
        self._size = min(self._size + steps, self._max_replay_buffer_size)

    # ---- sampling ----
    def _rows_per_batch(self, batch_size):
        assert batch_size % self.env_nums == 0, "batch size should be dividable by env_nums"
        return batch_size // self.env_nums

    def _gather(self, key, idx_dev, out=None):
        src = getattr(self, "_" + key)
        if out is not None:                                              # caller-owned (B, feat) destination
            if out.dtype != src.dtype or out.numel() != idx_dev.numel() * src[0].numel():
                raise _C.TrlError("random_batch: out[%r] does not match the batch" % key)
            _C.gather_rows(src, idx_dev, out=out.view((idx_dev.numel(),) + tuple(src.shape[1:])))
            return out
        block = _C.gather_rows(src, idx_dev)
        return block.reshape((block.shape[0] * self.env_nums,) + tuple(block.shape[2:]))

    def random_batch(self, batch_size, sample_key, out=None):
        """`out` (not in the reference): dict of preallocated (B, feat) device tensors to gather into, so that a
        captured update graph can read its inputs at fixed addresses; keys missing from it are allocated."""
        nrows = self._rows_per_batch(batch_size)
        indices = np.random.randint(0, self.num_steps_can_sample(), nrows)
        idx_dev = torch.from_numpy(indices.astype(np.int64)).to(self._device(), non_blocking=True)
        out = out or {}
        plain = [k for k in sample_key if self._plain_key(k)]
        if len(plain) < 2 or len(plain) > 8:
            return {key: self._gather(key, idx_dev, out.get(key)) for key in sample_key}
        # every plainly stored key in ONE launch (trl_gather_rows_multi); keys a subclass stores differently go alone
        batch, srcs, dsts = {}, [], []
        for key in plain:
            src, dst = getattr(self, "_" + key), out.get(key)
            if dst is None:
                dst = torch.empty((nrows * self.env_nums,) + tuple(src.shape[2:]), dtype=src.dtype, device=src.device)
            elif dst.dtype != src.dtype or dst.numel() != nrows * src[0].numel():
                raise _C.TrlError("random_batch: out[%r] does not match the batch" % key)
            batch[key] = dst
            srcs.append(src); dsts.append(dst)
        _C.gather_rows_multi(srcs, idx_dev, dsts)
        for key in sample_key:
            if key not in batch:
                batch[key] = self._gather(key, idx_dev, out.get(key))
        return {key: batch[key] for key in sample_key}

    def draw_indices(self, batch_size, count):
        """The row indices `count` successive `random_batch(batch_size, ...)` calls would draw, in that order from the same
        global numpy stream: (count, batch_size // env_nums) int64 on the host (not in the reference)."""
        # (one call: the legacy generator fills a (count, nrows) request element by element in C order, the values and the
        # generator state afterwards are those of `count` calls of size nrows -- tests/test_host_logic_cpu.py)
        return np.random.randint(0, self.num_steps_can_sample(), (count, self._rows_per_batch(batch_size))).astype(np.int64)

    def gather_sources(self, sample_key):
        """The stored (rows, N, ...) tensors of `sample_key` when every key is gathered by time row as stored (an update
        engine can then gather a sample itself from inside its captured graph), else None."""
        if not (2 <= len(sample_key) <= 8) or not all(self._plain_key(k) and hasattr(self, "_" + k) for k in sample_key):
            return None
        return [getattr(self, "_" + k) for k in sample_key]

    def _plain_key(self, key):
        """True when `_key[rows, N, ...]` is gathered by time row as stored (subclasses override for keys they keep
        in another form)."""
        return True

    def num_steps_can_sample(self):
        return self._size
