"""Frame-deduplicating replay buffer for stacked-frame observations -- the device counterpart of the reference's
MemoryEfficientReplayBuffer + LazyFrames (torchrl/replay_buffers/memory_efficient_replay_buffer.py:5-33,
torchrl/env/atari_wrapper.py:142-227): every 84x84 frame is stored once, the k-stacks of `obs` and `next_obs`
are rebuilt when a batch is drawn.

Layout: per env a ring of S single frames (`_stream[S, N, H*W]` uint8) plus, per replay row, the stream position
of the newest frame of `obs` (`_pos[rows, N]` int32); scalar keys (acts, rewards, terminals, time_limits) are
ordinary time-major tensors.  A step appends the one new frame of `next_obs`; an episode (re)start appends the
fresh stack's C frames.  S is sized from the ring length and the shortest episode the caller expects:
    S = rows + C * (rows // min_episode_frames + 2) + C
(one frame per transition instead of 2 C: 8x less HBM for C = 4).  If episodes turn out shorter, the gather
kernel raises a device flag instead of returning overwritten frames; `check_overrun()` reads it.

Same interface as BaseReplayBuffer: `random_batch` draws its row indices with the legacy numpy RNG
(bit-exact stream) and returns uint8 `(B, C, H, W)` stacks identical to what the plain buffer would hold.
The collector drives it through `begin_episodes` / `append_step` (torchrl_amd/collector/base.py).
"""
import torch

from .. import _C
from .base import BaseReplayBuffer


class MemoryEfficientReplayBuffer(BaseReplayBuffer):
    FRAME_KEYS = ("obs", "next_obs")

    def __init__(self, max_replay_buffer_size, env_nums=1, time_limit_filter=False, device=None,
                 min_episode_frames=None):
        super().__init__(max_replay_buffer_size, env_nums=env_nums, time_limit_filter=time_limit_filter, device=device)
        self.min_episode_frames = min_episode_frames
        self._stream = None

    # ---- stream management (called by the collector) ----
    def _ensure_stream(self, frame_shape):
        if self._stream is not None:
            return
        Cc, H, W = (int(v) for v in frame_shape)
        rows, N = self._max_replay_buffer_size, self.env_nums
        min_ep = max(1, int(self.min_episode_frames)) if self.min_episode_frames else 1
        S = rows + Cc * (rows // min_ep + 2) + Cc if self.min_episode_frames else 2 * rows + 3 * Cc
        dev = self._device()
        self.frame_shape = (Cc, H, W)
        self._stream = torch.zeros((S, N, H * W), dtype=torch.uint8, device=dev)
        self._head = torch.full((N,), -1, dtype=torch.int32, device=dev)       # nothing appended yet
        self._pos = torch.zeros((rows, N), dtype=torch.int32, device=dev)
        self._overrun = torch.zeros(1, dtype=torch.int32, device=dev)

    def begin_episodes(self, stacks, mask=None):
        """Append the whole stack of the envs in `mask` (all when None): call after env.reset() / a partial reset."""
        self._ensure_stream(stacks.shape[1:])
        _C.frame_stream_append(stacks, self._stream, self._head, mask, self.frame_shape[0])

    def mark_obs_row(self):
        """Record, for the row about to be written, where the current observation's newest frame lives."""
        self._pos[self._top].copy_(self._head)

    def append_step(self, stacks):
        """Append the newest frame of the post-step stacks (= the new frame of next_obs)."""
        _C.frame_stream_append(stacks, self._stream, self._head, None, 1)

    # ---- sampling ----
    def _plain_key(self, key):
        return key not in self.FRAME_KEYS

    def _gather(self, key, idx_dev, out=None):
        if key not in self.FRAME_KEYS:
            return super()._gather(key, idx_dev, out)
        return _C.frame_stream_gather(self._stream, self._pos, idx_dev, 0 if key == "obs" else 1, self.frame_shape,
                                      self._head, self._overrun, out=out)

    def check_overrun(self):
        """Raise if any batch asked for a frame that had already been overwritten (episodes shorter than
        `min_episode_frames` promised).  One small D2H; call at epoch boundaries."""
        if self._stream is not None and int(self._overrun.item()) != 0:
            raise _C.TrlError("frame stream overrun: episodes were shorter than min_episode_frames=%r; "
                              "construct the buffer with a smaller value" % (self.min_episode_frames,))

    def footprint_bytes(self):
        tot = 0 if self._stream is None else self._stream.numel() + 4 * (self._pos.numel() + self._head.numel())
        for k in self._keys:
            t = getattr(self, "_" + k)
            tot += t.numel() * t.element_size()
        return tot

    def add_sample(self, sample_dict, **kwargs):
        raise _C.TrlError("MemoryEfficientReplayBuffer is filled by the device collector (begin_episodes / append_step); "
                          "use BaseReplayBuffer for host-side add_sample")
