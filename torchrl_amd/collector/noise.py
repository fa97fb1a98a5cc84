"""The reference's exploration-noise stream, drawn by several host threads at once -- and by several ranks.

The reference samples `Normal(zeros, ones).sample()` from the CPU torch generator per vector step
(torchrl/policies/distribution.py:60-76) -- bit-identical to `torch.randn` on that generator.  One (T, N, A) block of
cfg 2 is 1.57 M values: ~3.2 ms of ONE host thread (the MT19937 engine is sequential), more than a whole PPO iteration
takes on the device.  The values do not have to be produced sequentially, though:

  * a float32 `normal_()` of n elements (n >= 16, n % 16 == 0) makes exactly n engine calls, element i from call i, and
    transforms the uniforms in aligned groups of 16 -- so a block cut at multiples of 16 is the concatenation of
    independent draws, each started from the engine state at its first element;
  * that state is obtained without drawing: `trl_mt19937_states_at` (include/trl_hip.h, host code) moves an engine state
    forward to any list of call positions at the cost of the 624-word twist per 624 calls (~0.3 ms per 1.5 M calls).

`draw_block(state0, out, ...)` derives the chunk states from `state0`, lets P threads run
`torch.randn(..., generator=g_p, out=chunk)` on PRIVATE generators (the op releases the interpreter lock) and returns the
state at the end of the block.  It never touches the default generator, so it can run on any thread; `randn_into(out)` is
the main-thread wrapper that reads the default generator's state, draws, and sets the state the single call would leave.

Env shards on several ranks (`stride` / `offset`): step t of the reference draws ONE (N_total, A) tensor for all envs
(distribution.py:60-76); the rank that owns envs [e0, e0 + N_local) needs elements [e0 * A, (e0 + N_local) * A) of it,
i.e. the T chunks at calls t * N_total * A + e0 * A of the stream, and the state after T * N_total * A calls at the end.
Every rank makes the same pass over the stream and draws only its own 1 / world of the values
(torchrl/replay_buffers/on_policy.py:75-88 is why column blocks of the global tensor are the right partition).

Nothing here is trusted blindly: the layout of torch's generator state and the block structure of its normal transform
are private to torch, so the first use runs a self-check against plain `torch.randn` calls (`fast_path_ok()`); when it
fails -- another torch build -- every caller falls back to the plain per-step draws and a warning is logged once.
"""
import ctypes as C
import logging
import os
import struct
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from .. import _C

# layout of torch's CPU generator state (ATen/CPUGeneratorImpl.cpp, CPUGeneratorImplState): uint64 seed | int32 left |
# int32 seeded | uint64 next | uint64 state[624] | normal_x, normal_y, normal_rho (double) | int32 normal_is_valid | pad |
# float next_float_normal_sample | bool valid | pad = 5056 bytes
_STATE_BYTES, _OFF_LEFT, _OFF_NEXT, _OFF_MT, _MT_N = 5056, 8, 16, 24, 624
MIN_PARALLEL = 1 << 16                       # below this a contiguous block is drawn by the calling thread
_GROUP_CHUNKS = 16                           # chunk states are derived this many at a time, the draws of a group run meanwhile

_pool, _pool_lock, _gens = None, threading.Lock(), threading.local()
_checked = None                              # None: not yet; True / False: result of the self-check
_log = logging.getLogger("torchrl_amd")
JUMP_MIN_CALLS = 1 << 21                     # streams at least this long are walked by several threads (jump-ahead)
STATS = {"blocks": 0, "parallel_blocks": 0, "pieces": 0, "native_blocks": 0, "jump_passes": 0}     # what draw_block did so far (tests assert the parallel path ran)


def default_threads():
    """Host threads of one draw: half of the CPUs THIS process may run on (a rank pinned to its slice of the host by
    dist.pin_rank_cpus must not size its pool -- or the jump-ahead threads of trl_mt19937_states_at_mt -- by the whole machine)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    return max(1, min(8, cores // 2))


def _executor(threads):
    global _pool
    with _pool_lock:
        if _pool is None or _pool._max_workers < threads:
            _pool = ThreadPoolExecutor(max_workers=threads, thread_name_prefix="trl-noise")
        return _pool


def _layout_known(state):
    return state.dtype == torch.uint8 and state.numel() == _STATE_BYTES


def states_at(state, positions, threads=1):
    """uint8 tensor (K, state bytes): `state` moved forward to each of the ascending engine-call positions.  threads > 1:
    the pass over a long stream is shared by that many host threads, all but the first starting from a JUMPED state
    (trl_mt19937_states_at_mt: the same records, byte for byte)."""
    if not _layout_known(state):
        raise _C.TrlError("CPU generator state of %d bytes: layout unknown to torchrl_amd.collector.noise" % state.numel())
    pos = np.ascontiguousarray(positions, dtype=np.int64)
    tmpl = state.contiguous()
    out = torch.empty(len(pos), _STATE_BYTES, dtype=torch.uint8)
    if threads > 1 and len(pos) > 1 and int(pos[-1]) >= JUMP_MIN_CALLS and os.environ.get("TRL_NOISE_JUMP") != "0":
        STATS["jump_passes"] += 1
        _C.check(_C.lib().trl_mt19937_states_at_mt(tmpl.data_ptr(), _STATE_BYTES, _OFF_LEFT, _OFF_NEXT, _OFF_MT,
                                                   pos.ctypes.data_as(C.c_void_p), len(pos), out.data_ptr(), int(threads)),
                 "trl_mt19937_states_at_mt")
        return out
    _C.check(_C.lib().trl_mt19937_states_at(tmpl.data_ptr(), _STATE_BYTES, _OFF_LEFT, _OFF_NEXT, _OFF_MT,
                                            pos.ctypes.data_as(C.c_void_p), len(pos), out.data_ptr()),
             "trl_mt19937_states_at")
    return out


def segment_states(state, n, parts):
    """Engine states at the P segment starts of an n-element block and at its end; returns (bounds, states) with
    bounds[p] .. bounds[p + 1] the elements of segment p (every bound a multiple of 16) and len(states) == P + 1."""
    seg = -(-n // parts)
    seg = (seg + 15) // 16 * 16
    bounds = list(range(0, n, seg)) + [n]
    recs = states_at(state, bounds)
    return bounds, [recs[k] for k in range(len(bounds))]


_ext_lib = False                             # False: not looked for yet; None: not available


def native_helper():
    """libtrl_noise.so (csrc/trl_noise_ext.cpp: torch's own normal_() on private generators from plain threads), or None."""
    global _ext_lib
    if _ext_lib is False:
        _ext_lib = None
        path = os.path.join(os.path.dirname(_C.LIB_PATH), "libtrl_noise.so")
        if os.environ.get("TRL_NOISE_HELPER") != "0" and os.path.exists(path):
            try:
                from .. import build as _build
                made_for, now = _build.noise_helper_built_for(path), _build.noise_helper_tag()
                if made_for != now:                  # built against another torch: loading it would map a second libtorch
                    _log.warning("libtrl_noise.so was built for torch %s, this is %s: not loaded (python -c 'import "
                                 "__graft_entry__ as g; g.build()' rebuilds it); the chunks are drawn from Python threads",
                                 made_for, now)
                    return None
                lib = C.CDLL(path)
                lib.trl_noise_draw_chunks.restype = C.c_int
                lib.trl_noise_draw_chunks.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
                lib.trl_noise_last_error.restype = C.c_char_p
                if lib.trl_noise_abi_version() == 1 and _native_check(lib):
                    _ext_lib = lib
            except (OSError, AttributeError):
                _ext_lib = None
    return _ext_lib


def _native_check(lib):
    """One block through the helper against plain `torch.randn` on a private generator, before it is trusted."""
    try:
        g = torch.Generator()
        g.manual_seed(977)
        torch.randn(100, generator=g)
        s0 = g.get_state()
        if not _layout_known(s0):
            return False
        n, cut = 4096, [0, 1024, 2048 + 16, 4096]
        want = torch.randn(n, generator=g)
        recs = states_at(s0, cut)
        got = torch.empty(n)
        _draw_native(lib, recs, got, [(a, a, b) for a, b in zip(cut[:-1], cut[1:])], 3)
        ok = bool(torch.equal(got, want))
        if not ok:
            _log.warning("libtrl_noise.so does not reproduce torch.randn on this build: not used")
        return ok
    except Exception:                                                   # noqa: BLE001
        return False


def _draw_native(lib, recs, flat, pieces, threads):
    off = np.ascontiguousarray([a for _, a, _ in pieces], dtype=np.int64)
    ln = np.ascontiguousarray([b - a for _, a, b in pieces], dtype=np.int64)
    rc = lib.trl_noise_draw_chunks(recs.data_ptr(), _STATE_BYTES, len(pieces), flat.data_ptr(), off.ctypes.data_as(C.c_void_p),
                                   ln.ctypes.data_as(C.c_void_p), int(threads))
    if rc != 0:
        raise _C.TrlError("trl_noise_draw_chunks failed (%d): %s" % (rc, lib.trl_noise_last_error().decode()))


def _draw(state, out):
    g = getattr(_gens, "g", None)
    if g is None:
        g = _gens.g = torch.Generator()
    g.set_state(state.clone())              # (a row VIEW with a storage offset crashes Generator.set_state on torch 2.10)
    torch.randn(out.numel(), generator=g, out=out)


def _draw_many(recs, outs):
    for st, o in zip(recs, outs):
        _draw(st, o)


def _self_check():
    """The three private facts this module rests on, checked against plain `torch.randn` calls on a private generator:
    the state layout (a derived state reproduces the engine), one call per element with aligned groups of 16 (a block
    equals its segments drawn separately), and one (T * n) draw == T successive n-element draws."""
    try:
        g = torch.Generator()
        g.manual_seed(20240917)
        torch.randn(37, generator=g)                                    # somewhere inside a 624-word block
        s0 = g.get_state()
        if not _layout_known(s0):
            return False
        n, cut = 2048, [0, 624 - 32, 1248, 2048]
        want = torch.randn(n, generator=g)
        end = g.get_state()
        recs = states_at(s0, cut)
        got = torch.empty(n)
        for a, b, st in zip(cut[:-1], cut[1:], recs):
            _draw(st, got[a:b])
        if not torch.equal(got, want) or not torch.equal(recs[-1], end):
            return False
        g.set_state(s0)
        steps = torch.cat([torch.randn(32 * 4, generator=g) for _ in range(16)])
        return bool(torch.equal(steps, want))
    except Exception:                                                   # noqa: BLE001 -- any surprise = do not trust it
        return False


def fast_path_ok():
    """True when this torch build's CPU generator behaves as the derived-state draws assume (checked once)."""
    global _checked
    if _checked is None:
        _checked = _self_check()
        if not _checked:
            _log.warning("torchrl_amd.collector.noise: this torch build's CPU generator does not match the layout / block "
                         "structure the parallel reference-noise draw relies on; falling back to plain per-step torch.randn "
                         "calls (same values, one host thread, no prefetch)")
    return _checked


def draw_block(state0, out, n_chunks=1, stride=None, offset=0, threads=None):
    """Fill the flat float32 CPU tensor `out` (n_chunks * L elements) with chunk t = elements
    [t * stride + offset, t * stride + offset + L) of the normal_() stream that starts at generator state `state0`, and
    return the state after n_chunks * stride elements (stride defaults to L: one contiguous block).  L, stride and offset
    must be multiples of 16.  Does not touch the default generator; any thread may call it."""
    n = out.numel()
    flat = out.view(-1)
    if n_chunks < 1 or n % n_chunks:
        raise _C.TrlError("draw_block: %d elements do not split into %d chunks" % (n, n_chunks))
    L = n // n_chunks
    stride = L if stride is None else int(stride)
    if out.dtype != torch.float32 or not out.is_contiguous() or out.device.type != "cpu":
        raise _C.TrlError("draw_block: a contiguous float32 CPU tensor is required")
    if L % 16 or stride % 16 or offset % 16 or L < 16 or offset + L > stride:
        raise _C.TrlError("draw_block: chunk length %d / stride %d / offset %d must be multiples of 16 (chunk inside stride)"
                          % (L, stride, offset))
    threads = default_threads() if threads is None else max(1, int(threads))
    if stride == L:                                                     # contiguous: cut into one segment per thread
        parts = threads if n >= MIN_PARALLEL else 1
        seg = (-(-n // parts) + 15) // 16 * 16
        starts = list(range(0, n, seg))
        pieces = [(a, a, min(a + seg, n)) for a in starts]             # (stream position, out start, out end)
    else:
        pieces = [(t * stride + offset, t * L, (t + 1) * L) for t in range(n_chunks)]
    end_pos = n_chunks * stride
    STATS["blocks"] += 1
    STATS["pieces"] += len(pieces)
    if len(pieces) == 1 or threads == 1:
        recs = states_at(state0, [p[0] for p in pieces] + [end_pos])
        _draw_many(recs[:-1], [flat[a:b] for _, a, b in pieces])
        return recs[-1].clone()
    STATS["parallel_blocks"] += 1
    ext = native_helper()
    if ext is not None:                                                 # one pass over the stream, then every chunk natively
        recs = states_at(state0, [p[0] for p in pieces] + [end_pos], threads=threads)
        _draw_native(ext, recs, flat, pieces, threads)
        STATS["native_blocks"] += 1
        return recs[-1].clone()
    pool = _executor(threads)
    jobs, state, at = [], state0, 0
    # the pass over the stream is sequential; its states are handed out in groups so that the draws of one group run while
    # the next group's states are derived
    group = max(_GROUP_CHUNKS, -(-len(pieces) // 64))
    for g0 in range(0, len(pieces), group):
        part = pieces[g0:g0 + group]
        last = g0 + group >= len(pieces)
        pos = [p[0] - at for p in part] + [(end_pos if last else pieces[g0 + group][0]) - at]
        recs = states_at(state, pos)
        per = -(-len(part) // threads)
        for k in range(0, len(part), per):
            jobs.append(pool.submit(_draw_many, recs[k:k + per], [flat[a:b] for _, a, b in part[k:k + per]]))
        state, at = recs[-1], at + pos[-1]
    for j in jobs:
        j.result()
    return state.clone()


def randn_into(out, threads=None):
    """Fill the contiguous float32 CPU tensor `out` with what `torch.randn(out.shape)` would return from the default CPU
    generator, and leave that generator in the state the single call would leave it in.  Main thread only (it reads and
    sets the default generator)."""
    n = out.numel()
    flat = out.view(-1)
    threads = default_threads() if threads is None else int(threads)
    if out.dtype != torch.float32 or not out.is_contiguous() or out.device.type != "cpu":
        raise _C.TrlError("randn_into: a contiguous float32 CPU tensor is required")
    if threads <= 1 or n < MIN_PARALLEL or n % 16 != 0 or not fast_path_ok():
        torch.randn(n, out=flat)
        return out
    torch.set_rng_state(draw_block(torch.get_rng_state(), flat, threads=threads))
    return out


def shard_ok(n_local, n_total, e0, a_dim):
    """Can the rank that owns envs [e0, e0 + n_local) of n_total draw its rows of each step's (n_total, A) tensor on its
    own?  (Chunk, stride and offset multiples of 16, and the self-check passed.)"""
    L, S, off = n_local * a_dim, n_total * a_dim, e0 * a_dim
    return L >= 16 and L % 16 == 0 and S % 16 == 0 and off % 16 == 0 and off + L <= S and fast_path_ok()


def randn_shard_into(out, n_steps, n_local, n_total, e0, a_dim, threads=None):
    """`out` (n_steps, n_local, A): rows [e0, e0 + n_local) of n_steps successive `torch.randn(n_total, A)` draws from the
    default CPU generator, which is left where those n_steps draws would leave it.  Main thread only."""
    if not shard_ok(n_local, n_total, e0, a_dim):
        raise _C.TrlError("randn_shard_into: shard not aligned to the generator's blocks of 16")
    end = draw_block(torch.get_rng_state(), out.view(-1), n_chunks=n_steps, stride=n_total * a_dim, offset=e0 * a_dim,
                     threads=threads)
    torch.set_rng_state(end)
    return out
