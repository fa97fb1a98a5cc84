/* libtrl_noise.so -- optional host helper of the reference's exploration-noise stream.
 *
 * The reference draws `Normal(zeros, ones).sample()` of shape (N_total, A) from the CPU torch generator on every vector
 * step (torchrl/policies/distribution.py:60-76, called from torchrl/collector/on_policy.py:95).  torchrl_amd keeps that
 * stream bit for bit but produces it in CHUNKS, each started from the generator state at its position in the stream
 * (include/trl_hip.h: trl_mt19937_states_at): the segments of one rollout block, or -- envs sharded over ranks -- this
 * rank's rows of every step's tensor.  This library fills such chunks with torch's own normal_() on private generators
 * from plain threads (no interpreter lock, no Python call per chunk).  It is built from
 * torchrl_amd/csrc/trl_noise_ext.cpp against the interpreter's libtorch by torchrl_amd/build.py and loaded with ctypes
 * by torchrl_amd/collector/noise.py; when it is absent the same chunks are drawn from Python threads (same values).
 * Host code only; plain C ABI (no torch types in the signatures). */
#ifndef TRL_NOISE_H
#define TRL_NOISE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int trl_noise_abi_version(void);                 /* 1 */
const char* trl_noise_last_error(void);          /* thread-local text of the last failure */

/* chunk k (0 <= k < n_chunks): generator-state image states[k * state_bytes .. (k + 1) * state_bytes) (torch's CPU
 * generator state, as trl_mt19937_states_at writes it) -> out[out_off[k] .. out_off[k] + out_len[k]) float32 standard
 * normals, exactly what `torch.randn(out_len[k], generator=g)` returns for a generator in that state; up to `threads`
 * chunks at a time.  0 on success. */
int trl_noise_draw_chunks(const uint8_t* states, int64_t state_bytes, int64_t n_chunks, float* out,
                          const int64_t* out_off, const int64_t* out_len, int threads);

#ifdef __cplusplus
}
#endif
#endif /* TRL_NOISE_H */
