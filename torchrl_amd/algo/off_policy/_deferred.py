"""Device-side pieces of the deferred-update protocol of the off-policy engines (`update_deferred` / `resolve_updates` /
`update_epoch_deferred`, see OffRLAlgo.update_per_epoch): where the logged numbers of an update wait for their one
read-back per epoch, and the replay indices of a whole epoch as one upload.  Neither exists in the reference, whose
update returns Python floats one `.item()` at a time (torchrl/algo/off_policy/twin_sac_q.py:181-207, dqn.py:62-72)."""
import numpy as np
import torch


class _Ref:
    """Where a handle's statistics live: the ring, or the copy made of it before it wrapped."""
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t


class StatRing:
    """`slots` statistics blocks of `width` elements: update u's block is row u % slots, written on the device (by a launch
    of the update itself that reads the device-resident update count, or by a stream-ordered copy).  The tensor's address
    never changes -- captured graphs hold it; if `slots` unread updates are pending when more are launched, the pending
    handles are re-pointed at a copy.
    (Measured and dropped: the ring in pinned host memory written through its device-mapped address -- the read-back is
    then only a stream synchronisation, but an epoch's resolve took the same 55 / 90 us (DQN / SAC), and page-locked
    tensors that torch copies asynchronously carry allocator events whose query aborts a graph capture in progress.)"""

    def __init__(self, slots, width, dtype, device):
        self.t = torch.zeros(int(slots), int(width), dtype=dtype, device=device)
        # rows of the ring whose update has been launched but not read yet, and the row the next update will be filed into
        self._ref, self._unread, self._next = _Ref(self.t), set(), None

    @property
    def slots(self):
        return int(self.t.shape[0])

    def make_room(self, count):
        """Call BEFORE launching `count` more updates: if one of the rows they will be filed into still holds an unread
        update, the pending handles are re-pointed at a copy of the ring (stream-ordered: after every pending update)."""
        if not self._unread:
            return
        nxt = 0 if self._next is None else self._next
        if count >= self.slots or any((nxt + k) % self.slots in self._unread for k in range(count)):
            self._ref.t = self.t.clone()
            self._ref, self._unread = _Ref(self.t), set()

    def handles(self, first, count):
        """(ref, row) of updates first .. first + count - 1, just launched."""
        rows = [(first + k) % self.slots for k in range(count)]
        self._unread.update(rows)
        self._next = (first + count) % self.slots
        return [(self._ref, row) for row in rows]

    def read(self, pairs):
        """Host rows of the given (ref, row) pairs, stacked in their order: one D2H per ring (copy); the only host sync.
        Reading a handle twice is harmless; a row counts as free once its handle has been read."""
        if not pairs:
            return torch.zeros(0, int(self.t.shape[1]), dtype=self.t.dtype).numpy()
        for ref, row in pairs:
            if ref is self._ref:
                self._unread.discard(row)
        if all(ref is self._ref for ref, _ in pairs):                    # the usual case: one ring, nothing wrapped
            return self.t.cpu().numpy()[[row for _, row in pairs]]
        rows_of = {}
        for ref, row in pairs:
            rows_of.setdefault(id(ref), (ref, []))[1].append(row)
        host = {key: iter(ref.t.cpu().numpy()[rows]) for key, (ref, rows) in rows_of.items()}
        return np.stack([next(host[id(ref)]) for ref, _ in pairs])


class IndexSlab:
    """The index sets of an epoch's `count` replay samples as ONE upload: {first update count, count, idx[count][nrows]}
    (int64) at a fixed device address per (count, nrows); trl_gather_rows_multi_dyn picks the set of the device-resident
    update count."""

    def __init__(self, device):
        self.device, self._bufs, self._stagers = device, {}, {}

    def upload(self, first, index_sets):
        from ... import _C
        count, nrows = index_sets.shape
        key = (count, nrows)
        if key not in self._bufs:
            self._bufs[key] = torch.zeros(2 + count * nrows, dtype=torch.int64, device=self.device)
            self._stagers[key] = _C.PinnedStager(2 + count * nrows, torch.int64)
        slab = self._bufs[key]
        # page-locked staging buffers in rotation, asynchronous copy: the host never waits for what is queued on the stream
        # (the collection just launched) before it may launch the update, and a host that runs epochs ahead of the device
        # does not overwrite a slab whose copy has not been executed yet (_C.PinnedStager)
        host = self._stagers[key].stage().numpy()
        host[0], host[1] = int(first), count
        host[2:] = index_sets.reshape(-1)
        return self._stagers[key].upload(slab)
