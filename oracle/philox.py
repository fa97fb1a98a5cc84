"""Philox4x32-10 counter-based generator + Box-Muller, numpy restatement.

Not part of the reference (which seeds gym envs with ``seed*N+i``,
torchrl/env/vecenv.py:63-65).  The synthetic benchmark env needs reset
observations / exploration noise that the CPU oracle and the HIP kernels can
both produce for env ``i`` regardless of evaluation order, so both sides
implement this published algorithm (Salmon et al., SC'11) with the same
key/counter convention:

    key     = (lo32(env_seed), hi32(env_seed))
    counter = (c0, c1, block, tag)

``tag`` separates streams (RESET / NOISE).  Uniforms are mapped to (0, 1) as
``(x >> 8) * 2**-24 + 2**-25`` (one fp32 rounding) and turned into normals by
Box-Muller on pairs: ``r = sqrt(-2 ln u0)``, ``z0 = r cos(2 pi u1)``,
``z1 = r sin(2 pi u1)``.  The HIP side does the same in fp32
(torchrl_amd/csrc/trl_philox.h); results agree to ~1e-6.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)

TAG_RESET = 0x52535421
TAG_NOISE = 0x4E4F4953


def philox4x32_10(ctr, key):
    """ctr: (..., 4) uint32 array-like, key: (..., 2) uint32. Returns (..., 4) uint32."""
    c = [np.asarray(ctr[..., i], dtype=np.uint64) for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint64)
    k1 = np.asarray(key[..., 1], dtype=np.uint64)
    for _ in range(10):
        p0 = M0 * c[0]
        p1 = M1 * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c = [(hi1 ^ c[1] ^ k0) & MASK, lo1, (hi0 ^ c[3] ^ k1) & MASK, lo0]
        k0 = (k0 + np.uint64(W0)) & MASK
        k1 = (k1 + np.uint64(W1)) & MASK
    return np.stack(c, axis=-1).astype(np.uint32)


def u32_to_unit(x):
    """uint32 -> float32 in (0, 1]; matches fmaf(float(x>>8), 2^-24, 2^-25)."""
    f = (np.asarray(x, dtype=np.uint32) >> np.uint32(8)).astype(np.float64)
    return (f * 2.0 ** -24 + 2.0 ** -25).astype(np.float32)


def normals4(c0, c1, block, tag, env_seed):
    """Four N(0,1) float32 per (broadcast) counter; env_seed is int64."""
    c0, c1, block, env_seed = np.broadcast_arrays(
        np.asarray(c0, dtype=np.int64), np.asarray(c1, dtype=np.int64),
        np.asarray(block, dtype=np.int64), np.asarray(env_seed, dtype=np.int64))
    ctr = np.stack([c0 & 0xFFFFFFFF, c1 & 0xFFFFFFFF, block & 0xFFFFFFFF,
                    np.full_like(c0, tag)], axis=-1).astype(np.uint32)
    key = np.stack([env_seed & 0xFFFFFFFF, (env_seed >> 32) & 0xFFFFFFFF],
                   axis=-1).astype(np.uint32)
    x = philox4x32_10(ctr, key)
    u = u32_to_unit(x).astype(np.float64)
    out = np.empty(u.shape, dtype=np.float64)
    for p in (0, 2):
        r = np.sqrt(-2.0 * np.log(u[..., p]))
        out[..., p] = r * np.cos(2.0 * np.pi * u[..., p + 1])
        out[..., p + 1] = r * np.sin(2.0 * np.pi * u[..., p + 1])
    return out.astype(np.float32)


def normal_vector(n, c0, c1, tag, env_seed):
    """(..., n) float32 normals: blocks 0..ceil(n/4)-1 concatenated, truncated to n."""
    nblk = (n + 3) // 4
    parts = [normals4(c0, c1, b, tag, env_seed) for b in range(nblk)]
    return np.concatenate(parts, axis=-1)[..., :n]
