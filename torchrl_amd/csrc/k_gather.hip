// K5/K6 -- row gather (minibatch slices / uniform replay sample) and
// K7 -- per-minibatch advantage statistics.
//
// A "row" is one time step of the (rows, N, feat) ring: N*feat contiguous
// elements, so the gather is a set of large contiguous copies selected by a
// host-generated int64 index (bit-exact with numpy's legacy RNG, see
// torchrl/replay_buffers/base.py:44, on_policy.py:76-78).  16-byte vector
// loads/stores when the row size allows; HBM-bound: row_bytes read + written.
#include <algorithm>
#include "trl_common.h"

template <typename VecT>
__global__ __launch_bounds__(256) void gather_rows_kernel(const VecT* __restrict__ src,
                                                          const int64_t* __restrict__ idx,
                                                          VecT* __restrict__ dst, int64_t row_vecs,
                                                          int64_t src_rows) {
  const int row = blockIdx.y;
  const int64_t s = idx[row];
  if (s < 0 || s >= src_rows) return;     // out-of-range index: host validates, never copy wild memory
  const VecT* sp = src + s * row_vecs;
  VecT* dp = dst + (int64_t)row * row_vecs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < row_vecs;
       i += (int64_t)gridDim.x * blockDim.x)
    dp[i] = sp[i];
}

template <typename VecT>
static int launch_gather(const void* src, const int64_t* idx, void* dst, int n_rows, int64_t row_vecs,
                         int64_t src_rows, hipStream_t s) {
  // enough workgroups to fill the chip whatever the row count: a cfg 5 sample is ONE 14 MB row (B = env_nums), which
  // 64 workgroups copied at 1 TB/s
  int bx = (int)((row_vecs + 255) / 256);
  const int cap = std::max(64, (4096 + n_rows - 1) / std::max(n_rows, 1));
  if (bx > cap) bx = cap;
  if (bx < 1) bx = 1;
  dim3 grid(bx, n_rows), block(256);
  hipLaunchKernelGGL(gather_rows_kernel<VecT>, grid, block, 0, s, (const VecT*)src, idx, (VecT*)dst,
                     row_vecs, src_rows);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

static int gather_bytes(const void* src, const int64_t* idx, void* dst, int n_rows, int64_t row_bytes,
                        int64_t src_rows, void* stream) {
  if (n_rows < 0 || row_bytes < 0 || src_rows < 0) { trl_set_error("gather_rows: negative size"); return TRL_EINVAL; }
  if (n_rows == 0 || row_bytes == 0) return TRL_OK;
  if (!src || !idx || !dst) { trl_set_error("gather_rows: null pointer"); return TRL_EINVAL; }
  if (n_rows > 65535) { trl_set_error("gather_rows: n_rows > 65535"); return TRL_EINVAL; }
  hipStream_t s = (hipStream_t)stream;
  const uintptr_t al = (uintptr_t)src | (uintptr_t)dst | (uintptr_t)row_bytes;
  if ((al & 15) == 0) return launch_gather<uint4>(src, idx, dst, n_rows, row_bytes / 16, src_rows, s);
  if ((al & 3) == 0) return launch_gather<uint32_t>(src, idx, dst, n_rows, row_bytes / 4, src_rows, s);
  return launch_gather<uint8_t>(src, idx, dst, n_rows, row_bytes, src_rows, s);
}

// Several keys of one replay sample in ONE launch (the uniform sample gathers obs / next_obs / acts / rewards /
// terminals with the same row index: five ~3 us dependent launches otherwise).  blockIdx.z = key.
#define GATHER_MAX_KEYS 8
struct GatherSet { const uint8_t* src[GATHER_MAX_KEYS]; uint8_t* dst[GATHER_MAX_KEYS]; int64_t row_bytes[GATHER_MAX_KEYS]; };
// `step` (optional): idx is then a SLAB {first update count, row sets, idx[row sets][gridDim.y]} and the row set is picked by
// the device-resident update counter -- a captured graph of several updates gathers a different sample in each.
__global__ __launch_bounds__(256) void gather_rows_multi_kernel(GatherSet g, const int64_t* __restrict__ idx, int64_t src_rows,
                                                                const double* __restrict__ step) {
  const int row = blockIdx.y, key = blockIdx.z;
  if (step) {
    const int64_t set = (int64_t)step[0] - idx[0];
    if (set < 0 || set >= idx[1]) return;                              // outside the slab: nothing is copied
    idx += 2 + set * (int64_t)gridDim.y;
  }
  const int64_t s = idx[row], nb = g.row_bytes[key];
  if (s < 0 || s >= src_rows) return;
  const uint8_t* sp = g.src[key] + s * nb;
  uint8_t* dp = g.dst[key] + (int64_t)row * nb;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, dt = (int64_t)gridDim.x * blockDim.x;
  if ((((uintptr_t)sp | (uintptr_t)dp | (uintptr_t)nb) & 15) == 0) {
    for (int64_t i = t0; i < nb / 16; i += dt) reinterpret_cast<uint4*>(dp)[i] = reinterpret_cast<const uint4*>(sp)[i];
  } else if ((((uintptr_t)sp | (uintptr_t)dp | (uintptr_t)nb) & 3) == 0) {
    for (int64_t i = t0; i < nb / 4; i += dt) reinterpret_cast<uint32_t*>(dp)[i] = reinterpret_cast<const uint32_t*>(sp)[i];
  } else {
    for (int64_t i = t0; i < nb; i += dt) dp[i] = sp[i];
  }
}
static int gather_multi(const void* const* src, void* const* dst, const int64_t* row_bytes, int n_keys,
                        const int64_t* row_idx, int n_rows, int64_t src_rows, const double* step, void* stream) {
  TRL_REQUIRE(n_keys >= 1 && n_keys <= GATHER_MAX_KEYS, "gather_rows_multi: 1..8 keys");
  TRL_REQUIRE(n_rows >= 0 && src_rows >= 0 && n_rows <= 65535, "gather_rows_multi: bad row count");
  if (n_rows == 0) return TRL_OK;
  TRL_REQUIRE(src && dst && row_bytes && row_idx, "gather_rows_multi: null pointer");
  GatherSet g;
  int64_t widest = 0;
  for (int k = 0; k < n_keys; ++k) {
    TRL_REQUIRE(src[k] && dst[k] && row_bytes[k] > 0, "gather_rows_multi: null key / empty row");
    g.src[k] = (const uint8_t*)src[k]; g.dst[k] = (uint8_t*)dst[k]; g.row_bytes[k] = row_bytes[k];
    widest = std::max(widest, row_bytes[k]);
  }
  int bx = (int)((widest / 16 + 255) / 256);
  const int cap = std::max(64, (4096 + n_rows - 1) / n_rows);
  bx = std::min(std::max(bx, 1), cap);
  hipLaunchKernelGGL(gather_rows_multi_kernel, dim3(bx, n_rows, n_keys), dim3(256), 0, (hipStream_t)stream, g, row_idx,
                     src_rows, step);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}
extern "C" int trl_gather_rows_multi(const void* const* src, void* const* dst, const int64_t* row_bytes, int n_keys,
                                     const int64_t* row_idx, const double* update_count, int n_rows, int64_t src_rows,
                                     void* stream) {
  return gather_multi(src, dst, row_bytes, n_keys, row_idx, n_rows, src_rows, update_count, stream);
}

extern "C" int trl_gather_rows_f32(const float* src, const int64_t* row_idx, float* dst, int n_rows,
                                   int64_t row_elems, int64_t src_rows, void* stream) {
  return gather_bytes(src, row_idx, dst, n_rows, row_elems * 4, src_rows, stream);
}

extern "C" int trl_gather_rows_u8(const uint8_t* src, const int64_t* row_idx, uint8_t* dst, int n_rows,
                                  int64_t row_bytes, int64_t src_rows, void* stream) {
  return gather_bytes(src, row_idx, dst, n_rows, row_bytes, src_rows, stream);
}

// ---------------------------------------------------------------- K7
// Per minibatch: sum / sum-of-squares in fp64 (so the unbiased variance (sumsq - sum^2/n)/(n-1) of ppo.py:142,147 has
// no cancellation problem), max and -min of the advantages of its time rows.
// A minibatch is cut into S slices of rows (slice s takes rows s, s + S, ...), one workgroup each -- 40 minibatches are
// 320 workgroups instead of 40 walking 65 536 values apiece.  A slice leaves its four partials in the workspace
// (device-scope stores, acknowledged before the slice counts itself in); the slice that arrives last adds the S partials
// in slice order and writes the minibatch's row of raw_out.  The counter is monotone (never reset): S arrivals per
// launch.  Deterministic.
// Side jobs of the same launch (each optional): `zero` doubles set to 0 (the statistics block the updates file their
// numbers into) and up to 4 copies of 4-byte words (target_pf <- pf, ppo.py:28-29 / utils.py:23-26; the epoch's row
// indices and learning rates from page-locked host memory, which the device reads in place) -- launches and copy
// commands of their own before, ~5 us each in front of every epoch's updates.
#define STATS_THREADS 1024
#define STATS_MAX_SPLIT 8
struct AdvStats {
  const float* advs; const int64_t* row_idx; int rows_mb, N, S;
  double* raw_out; double* part; unsigned* count;          // part: [n_mb][S][4]; count: [n_mb]   (S == 1: unused)
  double* zero; int64_t zero_n;
  int n_jobs; uint32_t* job_dst[4]; const uint32_t* job_src[4]; int64_t job_words[4];
};
__global__ __launch_bounds__(STATS_THREADS) void adv_stats_kernel(AdvStats a) {
  __shared__ double s_red[4][STATS_THREADS / 64];
  __shared__ int s_last;
  const int mb = blockIdx.x / a.S, sp = blockIdx.x - mb * a.S;
  const int rows_mb = a.rows_mb, N = a.N;
  {                                                          // side jobs, strided over the whole grid
    const int64_t t = (int64_t)blockIdx.x * STATS_THREADS + threadIdx.x, nt = (int64_t)gridDim.x * STATS_THREADS;
    for (int64_t e = t; e < a.zero_n; e += nt) a.zero[e] = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < a.n_jobs)
        for (int64_t e = t; e < a.job_words[q]; e += nt) a.job_dst[q][e] = a.job_src[q][e];
  }
  double sum = 0.0, sq = 0.0;
  float mx = -INFINITY, mn = INFINITY;
  // Rows are walked in groups of 8 whose indices and elements are all requested before the first is consumed (two memory
  // round trips per group instead of two per row); a thread adds its values in ascending row order.
  for (int r0 = sp; r0 < rows_mb; r0 += 8 * a.S) {
    int64_t ridx[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) ridx[k] = (r0 + k * a.S < rows_mb) ? a.row_idx[(size_t)mb * rows_mb + r0 + k * a.S] : 0;
    for (int i0 = threadIdx.x; i0 < N; i0 += 2 * STATS_THREADS) {
      float v[8][2];
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int i = i0 + u * STATS_THREADS;
          v[k][u] = (r0 + k * a.S < rows_mb && i < N) ? a.advs[(size_t)ridx[k] * N + i] : 0.0f;
        }
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int i = i0 + u * STATS_THREADS;
          if (r0 + k * a.S < rows_mb && i < N) {
            const float x = v[k][u];
            sum += (double)x; sq += (double)x * (double)x;
            mx = fmaxf(mx, x); mn = fminf(mn, x);
          }
        }
    }
  }
  sum = wave_sum(sum); sq = wave_sum(sq);
  mx = wave_max(mx); mn = -wave_max(-mn);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { s_red[0][wave] = sum; s_red[1][wave] = sq; s_red[2][wave] = mx; s_red[3][wave] = -mn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double r0 = 0, r1 = 0, r2 = -INFINITY, r3 = -INFINITY;
    for (int w = 0; w < STATS_THREADS / 64; ++w) {
      r0 += s_red[0][w]; r1 += s_red[1][w]; r2 = fmax(r2, s_red[2][w]); r3 = fmax(r3, s_red[3][w]);
    }
    if (a.S == 1) {
      a.raw_out[mb * 4 + 0] = r0; a.raw_out[mb * 4 + 1] = r1; a.raw_out[mb * 4 + 2] = r2; a.raw_out[mb * 4 + 3] = r3;
    } else {
      double* p = a.part + ((size_t)mb * a.S + sp) * 4;
      __hip_atomic_store(p + 0, r0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p + 1, r1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p + 2, r2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p + 3, r3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // partials acknowledged before this slice counts itself in
      const unsigned before = __hip_atomic_fetch_add(a.count + mb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((before + 1u) % (unsigned)a.S == 0u) {             // the last of this launch's S slices: fold in slice order
        const double* q = a.part + (size_t)mb * a.S * 4;
        double v[STATS_MAX_SPLIT][4];
#pragma unroll
        for (int s = 0; s < STATS_MAX_SPLIT; ++s)
#pragma unroll
          for (int k = 0; k < 4; ++k)
            v[s][k] = s < a.S ? __hip_atomic_load(q + s * 4 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        double t0 = 0, t1 = 0, t2 = -INFINITY, t3 = -INFINITY;
#pragma unroll
        for (int s = 0; s < STATS_MAX_SPLIT; ++s)
          if (s < a.S) { t0 += v[s][0]; t1 += v[s][1]; t2 = fmax(t2, v[s][2]); t3 = fmax(t3, v[s][3]); }
        a.raw_out[mb * 4 + 0] = t0; a.raw_out[mb * 4 + 1] = t1; a.raw_out[mb * 4 + 2] = t2; a.raw_out[mb * 4 + 3] = t3;
      }
    }
  }
}

static int adv_stats_split(int n_mb, int rows_mb) {          // slices per minibatch: ~256+ workgroups, >= 1 row each
  int s = 1;
  while (s < STATS_MAX_SPLIT && n_mb * s < 256 && 2 * s <= rows_mb) s *= 2;
  return s;
}
extern "C" int64_t trl_ppo_epoch_prologue_workspace(int n_mb) {   // bytes; zero them once
  return n_mb <= 0 ? 0 : (int64_t)n_mb * (STATS_MAX_SPLIT * 4 * sizeof(double) + sizeof(unsigned)) + 16;
}
extern "C" int trl_ppo_epoch_prologue_f64(const float* advs, const int64_t* row_idx, int n_mb, int rows_mb, int N,
                                          double* raw_out, void* workspace, double* zero, int64_t zero_doubles,
                                          int n_copies, void* const* copy_dst, const void* const* copy_src,
                                          const int64_t* copy_words, void* stream) {
  if (n_mb < 0 || rows_mb < 0 || N < 0 || zero_doubles < 0 || n_copies < 0 || n_copies > 4) { trl_set_error("epoch_prologue: bad size"); return TRL_EINVAL; }
  if (n_mb == 0) return TRL_OK;
  if (!advs || !row_idx || !raw_out) { trl_set_error("epoch_prologue: null pointer"); return TRL_EINVAL; }
  if ((zero_doubles && !zero) || (n_copies && (!copy_dst || !copy_src || !copy_words))) { trl_set_error("epoch_prologue: null side-job pointer"); return TRL_EINVAL; }
  AdvStats a{};
  a.advs = advs; a.row_idx = row_idx; a.rows_mb = rows_mb; a.N = N; a.raw_out = raw_out;
  a.S = workspace ? adv_stats_split(n_mb, rows_mb) : 1;
  a.part = (double*)workspace;
  a.count = workspace ? (unsigned*)((char*)workspace + (size_t)n_mb * STATS_MAX_SPLIT * 4 * sizeof(double)) : nullptr;
  a.zero = zero; a.zero_n = zero_doubles; a.n_jobs = n_copies;
  for (int q = 0; q < n_copies; ++q) {
    if (copy_words[q] < 0 || (copy_words[q] && (!copy_dst[q] || !copy_src[q])) ||
        ((reinterpret_cast<uintptr_t>(copy_dst[q]) | reinterpret_cast<uintptr_t>(copy_src[q])) & 3)) {
      trl_set_error("epoch_prologue: copy %d: null / misaligned pointer or negative size", q); return TRL_EINVAL;
    }
    a.job_dst[q] = (uint32_t*)copy_dst[q]; a.job_src[q] = (const uint32_t*)copy_src[q]; a.job_words[q] = copy_words[q];
  }
  hipLaunchKernelGGL(adv_stats_kernel, dim3(n_mb * a.S), dim3(STATS_THREADS), 0, (hipStream_t)stream, a);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}
extern "C" int trl_adv_stats_f64(const float* advs, const int64_t* row_idx, int n_mb, int rows_mb, int N,
                                 double* raw_out, void* stream) {
  return trl_ppo_epoch_prologue_f64(advs, row_idx, n_mb, rows_mb, N, raw_out, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, stream);
}
