"""The reference's exploration-noise stream, drawn by several host threads at once.

The reference samples `Normal(zeros, ones).sample()` from the CPU torch generator per vector step
(torchrl/policies/distribution.py:60-76) -- bit-identical to `torch.randn` on that generator.  One (T, N, A) block of
cfg 2 is 1.57 M values: ~3.2 ms of ONE host thread (the MT19937 engine is sequential), more than a whole PPO iteration
takes on the device.  The values do not have to be produced sequentially, though:

  * a float32 `normal_()` of n elements (n >= 16, n % 16 == 0) makes exactly n engine calls, element i from call i, and
    transforms the uniforms in aligned groups of 16 -- so a block cut at multiples of 16 is the concatenation of
    independent draws, each started from the engine state at its first element;
  * that state is obtained without drawing: `trl_mt19937_advance` (include/trl_hip.h, host code) moves an engine state
    forward by k calls at the cost of the 624-word twist per 624 calls (~0.3 ms for the whole block).

`randn_into(out)` therefore snapshots the default generator's state, derives the P segment states, lets P threads run
`torch.randn(..., generator=g_p, out=segment_p)` on private generators (the op releases the interpreter lock), and sets
the default generator to the end-of-block state: same values in the same places as ONE `torch.randn` call, same
generator state afterwards (tests/test_host_logic_cpu.py::test_parallel_reference_noise_*), ~P times faster.
"""
import ctypes as C
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
MIN_PARALLEL = 1 << 16                       # below this a block is drawn by the calling thread

_pool, _pool_lock, _gens = None, threading.Lock(), threading.local()


def default_threads():
    cores = os.cpu_count() or 1
    return max(1, min(8, cores // 2))


def _executor(threads):
    global _pool
    with _pool_lock:
        if _pool is None or _pool._max_workers < threads:
            _pool = ThreadPoolExecutor(max_workers=threads, thread_name_prefix="trl-noise")
        return _pool


def _parse(state):
    raw = state.numpy().tobytes()
    if len(raw) != _STATE_BYTES:
        raise _C.TrlError("CPU generator state of %d bytes: layout unknown to torchrl_amd.collector.noise" % len(raw))
    left, = struct.unpack_from("<i", raw, _OFF_LEFT)
    nxt, = struct.unpack_from("<Q", raw, _OFF_NEXT)
    mt = np.frombuffer(raw, dtype=np.uint64, count=_MT_N, offset=_OFF_MT).astype(np.uint32)
    return raw, left, nxt, mt


def _pack(template, left, nxt, mt):
    buf = bytearray(template)
    struct.pack_into("<i", buf, _OFF_LEFT, int(left))
    struct.pack_into("<Q", buf, _OFF_NEXT, int(nxt))
    buf[_OFF_MT:_OFF_MT + 8 * _MT_N] = mt.astype(np.uint64).tobytes()
    return torch.frombuffer(buf, dtype=torch.uint8).clone()


def segment_states(state, n, parts):
    """Engine states at the P segment starts of an n-element block and at its end; returns (bounds, states) with
    bounds[p] .. bounds[p + 1] the elements of segment p (every bound a multiple of 16) and len(states) == P + 1."""
    raw, left, nxt, mt = _parse(state)
    seg = -(-n // parts)
    seg = (seg + 15) // 16 * 16
    bounds = list(range(0, n, seg)) + [n]
    lib = _C.lib()
    mt = np.ascontiguousarray(mt)
    c_left, c_next = C.c_int32(left), C.c_int64(nxt)
    states = [state.clone()]
    for a, b in zip(bounds[:-1], bounds[1:]):
        _C.check(lib.trl_mt19937_advance(mt.ctypes.data_as(C.c_void_p), C.byref(c_left), C.byref(c_next), b - a),
                 "trl_mt19937_advance")
        states.append(_pack(raw, c_left.value, c_next.value, mt))
    return bounds, states


def _draw(state, out):
    g = getattr(_gens, "g", None)
    if g is None:
        g = _gens.g = torch.Generator()
    g.set_state(state)
    torch.randn(out.numel(), generator=g, out=out)


def randn_into(out, threads=None):
    """Fill the contiguous float32 CPU tensor `out` with what `torch.randn(out.shape)` would return from the default CPU
    generator, and leave that generator in the state the single call would leave it in."""
    n = out.numel()
    flat = out.view(-1)
    threads = default_threads() if threads is None else int(threads)
    if out.dtype != torch.float32 or not out.is_contiguous() or out.device.type != "cpu":
        raise _C.TrlError("randn_into: a contiguous float32 CPU tensor is required")
    if threads <= 1 or n < MIN_PARALLEL or n % 16 != 0:
        torch.randn(n, out=flat)
        return out
    bounds, states = segment_states(torch.get_rng_state(), n, threads)
    pool = _executor(threads)
    jobs = [pool.submit(_draw, states[p], flat[bounds[p]:bounds[p + 1]]) for p in range(len(bounds) - 1)]
    for j in jobs:
        j.result()
    torch.set_rng_state(states[-1])
    return out
