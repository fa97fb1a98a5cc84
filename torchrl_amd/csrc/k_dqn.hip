// K14 / K15 / K17 -- DQN TD loss, QR-DQN quantile-Huber loss (forward + output gradient), greedy /
// epsilon-greedy action selection, and the synthetic Atari-shaped frame env.
//
// Reference arithmetic: torchrl/algo/off_policy/dqn.py:53-60, qrdqn.py:39-60,
// torchrl/algo/utils.py:5-13 (quantile_regression_loss, huber),
// torchrl/policies/discrete_policies.py:40-67, 86-89.
#include "trl_common.h"
#include "trl_philox.h"

#define DQ_THREADS 256

__device__ __forceinline__ double dq_block_sum(double v, double* smem) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  double r = 0.0;
  for (int w = 0; w < DQ_THREADS / 64; ++w) r += smem[w];
  return r;
}

// ---------------------------------------------------------------- K14
// q_s_a = Q(s)[a];  target = r + gamma (1 - d) max_a' Q'(s')[a'];  loss = mean (q_s_a - target)^2
// dq (B, A) is zero except 2 (q_s_a - target) / B at the taken action.
// sums (3 doubles): loss sum, q_s_a sum, reward sum.
// Actions come as int64 (`act`) or as the floats the replay buffer stores (`act_f`, exactly one is non-null); with `ring`
// the three sums are also filed into row (update count % slots) of a (slots, 3) ring, read back once per epoch.
struct DqRing { double* ring; const double* step; int slots; };
// The stored action of sample b as an index in [0, A): a NaN or out-of-range stored value (an uninitialised row, a host env
// that returned a float action) must not turn into an out-of-bounds read of q and write of dq inside a replayed graph --
// it is clamped (NaN -> 0).  The reference raises an IndexError there (dqn.py:54, `gather`); the host side validates the
// actions it stores.
__device__ __forceinline__ int dq_action(const int64_t* act, const float* act_f, int b, int A) {
  int at;
  if (act) { const int64_t a = act[b]; at = a < 0 ? 0 : (a >= A ? A - 1 : (int)a); }
  else { const float a = act_f[b]; at = (a >= 0.0f) ? (a < (float)A ? (int)a : A - 1) : 0; }
  return at;
}
__device__ __forceinline__ void dq_file(const DqRing& r, double a, double b, double c) {
  if (!r.ring) return;
  const int64_t u = (int64_t)r.step[0];                                // (the optimiser step of this update comes later)
  double* row = r.ring + 3 * (((u % r.slots) + r.slots) % r.slots);
  row[0] = a; row[1] = b; row[2] = c;
}
__global__ __launch_bounds__(DQ_THREADS) void dqn_td_kernel(const float* __restrict__ q, const int64_t* __restrict__ act,
                                                            const float* __restrict__ act_f,
                                                            const float* __restrict__ qn, const float* __restrict__ rew,
                                                            const float* __restrict__ term, float gamma, int B, int A,
                                                            float* __restrict__ dq, double* __restrict__ sums, DqRing ring) {
  __shared__ double smem[DQ_THREADS / 64];
  double sl = 0, sq = 0, sr = 0;
  const float inv_b = 1.0f / (float)B;
  for (int b = threadIdx.x; b < B; b += DQ_THREADS) {
    float mx = -INFINITY;
    for (int a = 0; a < A; ++a) mx = fmaxf(mx, qn[(size_t)b * A + a]);
    const int at = dq_action(act, act_f, b, A);
    const float qsa = q[(size_t)b * A + at];
    const float tgt = rew[b] + gamma * (1.0f - term[b]) * mx;
    const float e = qsa - tgt;
    for (int a = 0; a < A; ++a) dq[(size_t)b * A + a] = (a == at) ? 2.0f * e * inv_b : 0.0f;
    sl += (double)e * e; sq += (double)qsa; sr += (double)rew[b];
  }
  sl = dq_block_sum(sl, smem); sq = dq_block_sum(sq, smem); sr = dq_block_sum(sr, smem);
  if (threadIdx.x == 0) { sums[0] = sl; sums[1] = sq; sums[2] = sr; dq_file(ring, sl, sq, sr); }
}
static int dqn_td(const float* q, const int64_t* acts, const float* acts_f, const float* q_next, const float* rewards,
                  const float* terminals, float gamma, int B, int A, float* dq, double* sums, DqRing ring, void* stream) {
  TRL_REQUIRE(B > 0 && A > 0, "bad sizes");
  TRL_REQUIRE(q && (acts || acts_f) && q_next && rewards && terminals && dq && sums, "null pointer");
  hipLaunchKernelGGL(dqn_td_kernel, dim3(1), dim3(DQ_THREADS), 0, (hipStream_t)stream, q, acts, acts_f, q_next, rewards,
                     terminals, gamma, B, A, dq, sums, ring);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}
extern "C" int trl_dqn_td_loss_f32(const float* q, const int64_t* acts, const float* acts_f, const float* q_next,
                                   const float* rewards, const float* terminals, float gamma, int B, int A, float* dq,
                                   double* sums, double* ring, int slots, const double* update_count, void* stream) {
  TRL_REQUIRE(!acts != !acts_f, "dqn_td_loss: exactly one of acts (int64) / acts_f (float)");
  TRL_REQUIRE(!ring || (slots > 0 && update_count), "dqn_td_loss: ring without slots / counter");
  return dqn_td(q, acts, acts_f, q_next, rewards, terminals, gamma, B, A, dq, sums, DqRing{ring, update_count, slots}, stream);
}

// ---------------------------------------------------------------- K15
// theta = Q(s).view(B, A, Q)[a] (Q quantiles);  a* = argmax_a mean_i Q'(s')[a][i];
// T_i = r + gamma (1 - d) Q'(s')[a*][i];  diff[i][j] = T_i - theta_j;
// loss = mean_{b,i,j} huber(diff) * |tau_j - 1[diff < 0]|,  tau_j = (2j + 1) / 2Q.
// One workgroup per sample: T and theta staged in LDS, thread j owns theta_j's column of the
// Q x Q table (Q^2 = 40 000 Huber terms per sample at Q = 200 -- VALU-bound, 1.6 KB of input).
__global__ __launch_bounds__(DQ_THREADS) void quantile_huber_kernel(const float* __restrict__ q, const int64_t* __restrict__ act,
                                                                    const float* __restrict__ act_f,
                                                                    const float* __restrict__ qn, const float* __restrict__ rew,
                                                                    const float* __restrict__ term, float gamma, int B, int A,
                                                                    int Q, float* __restrict__ dq,
                                                                    double* __restrict__ part /* (B, 2): loss, q_s_a sums */) {
  extern __shared__ __attribute__((aligned(16))) float sm[];                     // T[Q] | theta[Q] | means[A]
  __shared__ double smem[DQ_THREADS / 64];
  __shared__ int s_astar;
  float* T = sm;
  float* th = sm + Q;
  float* means = sm + 2 * Q;
  const int b = blockIdx.x;
  const float* qb = q + (size_t)b * A * Q;
  const float* nb = qn + (size_t)b * A * Q;
  // greedy next action by mean over quantiles (first maximal index, like torch.max)
  for (int a = threadIdx.x >> 6; a < A; a += DQ_THREADS / 64) {
    float s = 0.0f;
    for (int i = threadIdx.x & 63; i < Q; i += 64) s += nb[a * Q + i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) means[a] = s / (float)Q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int best = 0;
    for (int a = 1; a < A; ++a) if (means[a] > means[best]) best = a;
    s_astar = best;
  }
  __syncthreads();
  const int at = dq_action(act, act_f, b, A), as = s_astar;
  const float r = rew[b], nd = gamma * (1.0f - term[b]);
  for (int i = threadIdx.x; i < Q; i += DQ_THREADS) { T[i] = r + nd * nb[as * Q + i]; th[i] = qb[at * Q + i]; }
  for (int e = threadIdx.x; e < A * Q; e += DQ_THREADS) dq[(size_t)b * A * Q + e] = 0.0f;
  __syncthreads();
  const float norm = 1.0f / ((float)B * (float)Q * (float)Q);
  double loss = 0.0, qsum = 0.0;
  for (int j = threadIdx.x; j < Q; j += DQ_THREADS) {
    const float tj = th[j], tau = (2.0f * j + 1.0f) / (2.0f * Q), one_m_tau = 1.0f - tau;
    float l = 0.0f, g = 0.0f;
    // Q^2 Huber terms per sample make this launch VALU-bound (40 000 per sample at Q = 200): nine vector instructions per
    // term instead of fourteen -- with m = min(|d|, 1): huber(d) = m (|d| - m / 2) (= d^2 / 2 inside, |d| - 1/2 outside),
    // huber'(d) = clamp(d, -1, 1) (one v_med3), weight = d < 0 ? 1 - tau : tau (one select)
    auto term = [&](float Ti) {
      const float d = Ti - tj;
      const float ad = fabsf(d);
      const float w = d < 0.0f ? one_m_tau : tau;
      const float m = fminf(ad, 1.0f);
      l = fmaf(m * fmaf(-0.5f, m, ad), w, l);
      g = fmaf(__builtin_amdgcn_fmed3f(d, -1.0f, 1.0f), w, g);  // d huber / d diff  (d diff / d theta_j = -1)
    };
    int i = 0;
    for (; i + 4 <= Q; i += 4) {                                 // T is read four values at a time (16-byte LDS reads)
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(T + i);
      term(t4[0]); term(t4[1]); term(t4[2]); term(t4[3]);
    }
    for (; i < Q; ++i) term(T[i]);
    dq[(size_t)b * A * Q + at * Q + j] = -g * norm;
    loss += (double)l; qsum += (double)tj;
  }
  loss = dq_block_sum(loss, smem); qsum = dq_block_sum(qsum, smem);
  if (threadIdx.x == 0) { part[b * 2 + 0] = loss; part[b * 2 + 1] = qsum; }
}
__global__ __launch_bounds__(DQ_THREADS) void quantile_fold_kernel(const double* __restrict__ part,
                                                                   const float* __restrict__ rew, int B,
                                                                   double* __restrict__ sums, DqRing ring) {
  __shared__ double smem[DQ_THREADS / 64];
  double a = 0, c = 0, r = 0;
  for (int b = threadIdx.x; b < B; b += DQ_THREADS) { a += part[b * 2]; c += part[b * 2 + 1]; r += (double)rew[b]; }
  a = dq_block_sum(a, smem); c = dq_block_sum(c, smem); r = dq_block_sum(r, smem);
  if (threadIdx.x == 0) { sums[0] = a; sums[1] = c; sums[2] = r; dq_file(ring, a, c, r); }
}
static int quantile_huber(const float* q, const int64_t* acts, const float* acts_f, const float* q_next,
                          const float* rewards, const float* terminals, float gamma, int B, int A, int Q, float* dq,
                          double* workspace, double* sums, DqRing ring, void* stream);
extern "C" int trl_quantile_huber_f32(const float* q, const int64_t* acts, const float* acts_f, const float* q_next,
                                      const float* rewards, const float* terminals, float gamma, int B, int A, int Q,
                                      float* dq, double* workspace /* 2B doubles */, double* sums, double* ring, int slots,
                                      const double* update_count, void* stream) {
  TRL_REQUIRE(!acts != !acts_f, "quantile_huber: exactly one of acts (int64) / acts_f (float)");
  TRL_REQUIRE(!ring || (slots > 0 && update_count), "quantile_huber: ring without slots / counter");
  return quantile_huber(q, acts, acts_f, q_next, rewards, terminals, gamma, B, A, Q, dq, workspace, sums,
                        DqRing{ring, update_count, slots}, stream);
}
static int quantile_huber(const float* q, const int64_t* acts, const float* acts_f, const float* q_next,
                          const float* rewards, const float* terminals, float gamma, int B, int A, int Q, float* dq,
                          double* workspace, double* sums, DqRing ring, void* stream) {
  TRL_REQUIRE(B > 0 && A > 0 && Q > 0 && A <= 64 && Q <= 4096, "bad sizes");
  TRL_REQUIRE(q && (acts || acts_f) && q_next && rewards && terminals && dq && workspace && sums, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(quantile_huber_kernel, dim3(B), dim3(DQ_THREADS), (2 * Q + A) * sizeof(float), s, q, acts, acts_f,
                     q_next, rewards, terminals, gamma, B, A, Q, dq, workspace);
  TRL_LAUNCH_CHECK();
  hipLaunchKernelGGL(quantile_fold_kernel, dim3(1), dim3(DQ_THREADS), 0, s, workspace, rewards, B, sums, ring);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// ---------------------------------------------------------------- K14b
// The linear head of a (non-quantile) DQN update as ONE launch: Q(s) = h W^T + b and Q'(s') = h' W'^T + b' on the last
// hidden activations of the online / target net, K14's loss and sums, and the head's whole backward pass
//     dh[b, :] = dq[b, a_b] W[a_b, :],   dW[a, :] = sum_b dq[b, a] h[b, :],   db[a] = sum_b dq[b, a]
// (dqn.py:47-60 + the autograd of nn.Linear).  With A = 6 outputs these were seven launches -- two split-K GEMMs with
// their folds, the loss, a weight-gradient GEMM of 8 workgroups with its fold, an input-gradient GEMM -- of 5-11 us each,
// every one latency-bound.  Here a wave owns a sample: the two rows of activations are 2 x H floats in registers, W and W'
// sit in LDS; the eight samples of a pass leave their activations and their loss factor in LDS, and every thread adds
// them -- in wave order -- into its own slice of the workgroup's dW; that partial leaves the workgroup as self-validating
// granules and each of the 64 workgroups folds a slice of the outputs over all partials in fixed order.  Deterministic.
#define DQH_WGS 64
#define DQH_WAVES 8
#define DQH_THREADS (64 * DQH_WAVES)
#define DQH_MAX_A 8
#define DQH_MAX_H 1024
#define DQH_MAX_GRP 4                             // float4 groups of dW per thread: A * H <= 8192
struct DqHead {
  const float* h; const float* hn; const float* w; const float* bias; const float* wt; const float* bias_t;
  const int64_t* act; const float* act_f; const float* rew; const float* term;
  float gamma; int B, H, A;
  float* dh; float* dw; float* db; float* q_out; float* qn_out; double* sums; DqRing ring;
  unsigned long long* part;                       // [DQH_WGS][A * H + A + 6] granules {launch epoch, value bits}
  unsigned long long* arrive;                     // launches so far x DQH_WGS: where the launch epoch comes from
};
#ifdef TRL_EXP_CLK                                // development aid (tools/bench_dqn_head.py): 100 MHz stamps of workgroups 0 and 63
__device__ long long g_dqh_clk[2 * 8];
#define HCLK(ph) if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == DQH_WGS - 1)) g_dqh_clk[(blockIdx.x ? 8 : 0) + (ph)] = wall_clock64();
extern "C" int trl_dbg_dqh_clk(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dqh_clk), sizeof(long long) * 16); }
#else
#define HCLK(ph)
#endif
// A partial leaves its workgroup as 8-byte granules {epoch, value}, written and read with device-scope accesses that
// bypass the (per-XCD, non-coherent) L2: a reader that sees this launch's epoch in a granule has its value.  No fence
// (a device-scope release / acquire pair writes back and invalidates a whole L2: 7 us each, measured) and no barrier.
__device__ __forceinline__ void dqh_put(unsigned long long* p, unsigned epoch, float v) {
  __hip_atomic_store(p, ((unsigned long long)epoch << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float dqh_get(const unsigned long long* p, unsigned epoch) {
  unsigned long long g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while ((unsigned)(g >> 32) != epoch) {
    __builtin_amdgcn_s_sleep(1);
    g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return __uint_as_float((unsigned)g);
}
template <int CH>                                 // chunks of 256 floats per activation row: H <= 256 CH
__global__ __launch_bounds__(DQH_THREADS) void dqn_head_kernel(DqHead d) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  __shared__ float swave[DQH_WAVES][6];           // loss / q_s_a / reward sums as (hi, lo) float pairs
  __shared__ float sbias[DQH_WAVES][DQH_MAX_A];
  __shared__ float comb[DQH_WAVES][64];
  __shared__ float hb[2][DQH_MAX_A];
  __shared__ float meta_g[DQH_WAVES];
  __shared__ int meta_at[DQH_WAVES];
  __shared__ unsigned s_epoch;
  const int H = d.H, A = d.A, AH = A * H, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  HCLK(0)
  if (tid == 0)                                   // launches are stream-ordered: all 64 arrivals of a launch share old / 64
    s_epoch = (unsigned)(__hip_atomic_fetch_add(d.arrive, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / DQH_WGS) + 1u;
  float* Ws = dsm;                                // [A][H]
  float* Wts = dsm + AH;                          // [A][H]
  float* rowbuf = dsm + 2 * AH;                   // [8 waves][H]: the activations of the samples of one pass
  for (int e = tid * 4; e < AH; e += DQH_THREADS * 4) {
    *reinterpret_cast<f32x4*>(Ws + e) = *reinterpret_cast<const f32x4*>(d.w + e);
    *reinterpret_cast<f32x4*>(Wts + e) = *reinterpret_cast<const f32x4*>(d.wt + e);
  }
  if (tid < DQH_MAX_A) hb[0][tid] = (d.bias && tid < A) ? d.bias[tid] : 0.0f;
  else if (tid < 2 * DQH_MAX_A) hb[1][tid - DQH_MAX_A] = (d.bias_t && tid - DQH_MAX_A < A) ? d.bias_t[tid - DQH_MAX_A] : 0.0f;
  const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
  // this THREAD's slice of the workgroup's dW: float4 groups tid + 512 k of the (A, H) matrix
  f32x4 gw[DQH_MAX_GRP];
#pragma unroll
  for (int k = 0; k < DQH_MAX_GRP; ++k) gw[k] = zero4;
  float gb[DQH_MAX_A];                            // this WAVE's db (wave-uniform)
#pragma unroll
  for (int a = 0; a < DQH_MAX_A; ++a) gb[a] = 0.0f;
  __syncthreads();
  HCLK(1)
  const unsigned epoch = s_epoch;
  const float inv_b = 1.0f / (float)d.B;
  double sl = 0.0, sq = 0.0, sr = 0.0;
  for (int b0 = blockIdx.x * DQH_WAVES; b0 < d.B; b0 += DQH_WGS * DQH_WAVES) {       // a pass: one sample per wave
    const int b = b0 + wave;
    float g = 0.0f;
    int at = 0;
    if (b < d.B) {
      f32x4 hv[CH], nv[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int j = c * 256 + lane * 4;
        hv[c] = j < H ? *reinterpret_cast<const f32x4*>(d.h + (size_t)b * H + j) : zero4;
        nv[c] = j < H ? *reinterpret_cast<const f32x4*>(d.hn + (size_t)b * H + j) : zero4;
      }
      at = dq_action(d.act, d.act_f, b, A);
      const float r = d.rew[b], nd = d.gamma * (1.0f - d.term[b]);
      // the 2 A dot products: lane partials first, then ONE butterfly over all of them (independent shuffles pipeline;
      // a reduction per dot product is 12 dependent chains of 6 LDS round trips)
      float s[DQH_MAX_A], sn[DQH_MAX_A];
#pragma unroll
      for (int a = 0; a < DQH_MAX_A; ++a) {
        s[a] = 0.0f; sn[a] = 0.0f;
        if (a < A) {
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            const int j = c * 256 + lane * 4;
            if (j < H) {
              const f32x4 wv = *reinterpret_cast<const f32x4*>(Ws + a * H + j), tv = *reinterpret_cast<const f32x4*>(Wts + a * H + j);
              s[a] = fmaf(hv[c][0], wv[0], s[a]); s[a] = fmaf(hv[c][1], wv[1], s[a]);
              s[a] = fmaf(hv[c][2], wv[2], s[a]); s[a] = fmaf(hv[c][3], wv[3], s[a]);
              sn[a] = fmaf(nv[c][0], tv[0], sn[a]); sn[a] = fmaf(nv[c][1], tv[1], sn[a]);
              sn[a] = fmaf(nv[c][2], tv[2], sn[a]); sn[a] = fmaf(nv[c][3], tv[3], sn[a]);
            }
          }
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int a = 0; a < DQH_MAX_A; ++a)
          if (a < A) { s[a] += __shfl_xor(s[a], o, 64); sn[a] += __shfl_xor(sn[a], o, 64); }
      }
      float mx = -INFINITY, qsa = 0.0f;
#pragma unroll
      for (int a = 0; a < DQH_MAX_A; ++a) {
        if (a < A) {
          const float qa = s[a] + hb[0][a], qna = sn[a] + hb[1][a];
          if (lane == 0 && d.q_out) d.q_out[(size_t)b * A + a] = qa;
          if (lane == 0 && d.qn_out) d.qn_out[(size_t)b * A + a] = qna;
          mx = fmaxf(mx, qna);
          qsa = a == at ? qa : qsa;
        }
      }
      const float tgt = r + nd * mx;
      const float e = qsa - tgt;
      g = 2.0f * e * inv_b;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int j = c * 256 + lane * 4;
        if (j < H) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(Ws + at * H + j);
          f32x4 o = {g * wv[0], g * wv[1], g * wv[2], g * wv[3]};
          *reinterpret_cast<f32x4*>(d.dh + (size_t)b * H + j) = o;
          *reinterpret_cast<f32x4*>(rowbuf + wave * H + j) = hv[c];
        }
      }
#pragma unroll
      for (int a = 0; a < DQH_MAX_A; ++a) gb[a] += a == at ? g : 0.0f;
      sl += (double)e * e; sq += (double)qsa; sr += (double)r;
    }
    if (lane == 0) { meta_g[wave] = g; meta_at[wave] = at; }
    __syncthreads();
    // dW += sum over the pass's samples, in wave order: row a_b of the matrix gets g_b h_b, the other rows nothing
#pragma unroll
    for (int k = 0; k < DQH_MAX_GRP; ++k) {
      const int e4 = 4 * (tid + DQH_THREADS * k);
      if (e4 < AH) {
        const int a = e4 / H, j = e4 - a * H;
#pragma unroll
        for (int w = 0; w < DQH_WAVES; ++w) {
          const float ga = meta_at[w] == a ? meta_g[w] : 0.0f;
          const f32x4 x = *reinterpret_cast<const f32x4*>(rowbuf + w * H + j);
          gw[k][0] = fmaf(ga, x[0], gw[k][0]); gw[k][1] = fmaf(ga, x[1], gw[k][1]);
          gw[k][2] = fmaf(ga, x[2], gw[k][2]); gw[k][3] = fmaf(ga, x[3], gw[k][3]);
        }
      }
    }
    __syncthreads();
  }
  HCLK(2)
  if (lane == 0) {
    // a double travels as two floats (hi + lo): the granules carry 32 value bits
    const float lh = (float)sl, qh = (float)sq, rh = (float)sr;
    swave[wave][0] = lh; swave[wave][1] = (float)(sl - (double)lh);
    swave[wave][2] = qh; swave[wave][3] = (float)(sq - (double)qh);
    swave[wave][4] = rh; swave[wave][5] = (float)(sr - (double)rh);
#pragma unroll
    for (int a = 0; a < DQH_MAX_A; ++a) sbias[wave][a] = gb[a];
  }
  const int n_out = AH + A, n_gran = n_out + 6;
  unsigned long long* part = d.part + (size_t)blockIdx.x * n_gran;
#pragma unroll
  for (int k = 0; k < DQH_MAX_GRP; ++k) {
    const int e4 = 4 * (tid + DQH_THREADS * k);
    if (e4 < AH) {
#pragma unroll
      for (int i = 0; i < 4; ++i) dqh_put(part + e4 + i, epoch, gw[k][i]);
    }
  }
  __syncthreads();
  HCLK(3)
  if (wave == 1 && lane < A) {
    float v = 0.0f;
    for (int w = 0; w < DQH_WAVES; ++w) v += sbias[w][lane];
    dqh_put(part + AH + lane, epoch, v);
  }
  if (wave == 2 && lane < 3) {                      // this workgroup's three sums, again as (hi, lo)
    double v = 0.0;
    for (int w = 0; w < DQH_WAVES; ++w) v += (double)swave[w][2 * lane] + (double)swave[w][2 * lane + 1];
    const float hi = (float)v;
    dqh_put(part + n_out + 2 * lane, epoch, hi);
    dqh_put(part + n_out + 2 * lane + 1, epoch, (float)(v - (double)hi));
  }
  HCLK(4)
  // fold over the workgroups: a pass = 64 outputs, wave w adds partials 8 w .. 8 w + 7 (all loads in flight), then the
  // eight sub-sums are added in order
  for (int e0 = blockIdx.x * 64; e0 < n_out; e0 += DQH_WGS * 64) {
    const int e = e0 + lane;
    float x[DQH_WGS / DQH_WAVES];
#pragma unroll
    for (int i = 0; i < DQH_WGS / DQH_WAVES; ++i)
      x[i] = e < n_out ? dqh_get(d.part + (size_t)(wave * (DQH_WGS / DQH_WAVES) + i) * n_gran + e, epoch) : 0.0f;
    float v = x[0];
#pragma unroll
    for (int i = 1; i < DQH_WGS / DQH_WAVES; ++i) v += x[i];
    comb[wave][lane] = v;
    __syncthreads();
    if (wave == 0 && e < n_out) {
      float t = comb[0][lane];
      for (int w = 1; w < DQH_WAVES; ++w) t += comb[w][lane];
      if (e < AH) d.dw[e] = t; else d.db[e - AH] = t;
    }
    __syncthreads();
  }
  HCLK(5)
  if (blockIdx.x == DQH_WGS - 1 && wave == DQH_WAVES - 1) {
    // the three sums: lane g fetches workgroup g's pairs; added in workgroup order by lane 0
    double mine[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const unsigned long long* q = d.part + (size_t)lane * n_gran + n_out + 2 * k;
      mine[k] = (double)dqh_get(q, epoch) + (double)dqh_get(q + 1, epoch);
    }
    double tot[3] = {0.0, 0.0, 0.0};
    for (int g = 0; g < DQH_WGS; ++g) {
#pragma unroll
      for (int k = 0; k < 3; ++k) tot[k] += __shfl(mine[k], g, 64);
    }
    if (lane == 0) {
      d.sums[0] = tot[0]; d.sums[1] = tot[1]; d.sums[2] = tot[2];
      dq_file(d.ring, tot[0], tot[1], tot[2]);
    }
  }
  HCLK(6)
}
template <int CH>
static int launch_dqn_head(const DqHead& d, int lds, hipStream_t s) {
  static int attr_lds = 0;
  if (lds > attr_lds) {
    hipError_t e = hipFuncSetAttribute((const void*)dqn_head_kernel<CH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { trl_set_error("dqn_head: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_lds = lds;
  }
  hipLaunchKernelGGL(dqn_head_kernel<CH>, dim3(DQH_WGS), dim3(DQH_THREADS), lds, s, d);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}
extern "C" int trl_dqn_head_supported(int H, int A) {
  return H >= 4 && (H & 3) == 0 && H <= DQH_MAX_H && A >= 1 && A <= DQH_MAX_A &&
         A * H <= 2048 * DQH_MAX_GRP;
}
extern "C" int64_t trl_dqn_head_workspace(int H, int A) {           // bytes; zero them once (the arrival counter lives there)
  if (!trl_dqn_head_supported(H, A)) return 0;
  return (int64_t)DQH_WGS * (A * H + A + 6) * sizeof(unsigned long long) + 16;
}
extern "C" int trl_dqn_head_f32(const float* h, const float* h_next, const float* w, const float* bias, const float* w_t,
                                const float* bias_t, const int64_t* acts, const float* acts_f, const float* rewards,
                                const float* terminals, float gamma, int B, int H, int A, float* dh, float* dw, float* db,
                                float* q_out, float* qn_out, double* sums, double* ring, int slots,
                                const double* update_count, void* workspace, void* stream) {
  TRL_REQUIRE(B > 0 && trl_dqn_head_supported(H, A), "dqn_head: H % 4 == 0, H <= 1024, A <= 8 (use the layer kernels + trl_dqn_td_loss_f32)");
  TRL_REQUIRE(h && h_next && w && w_t && rewards && terminals && dh && dw && db && sums && workspace, "null pointer");
  TRL_REQUIRE(!acts != !acts_f, "dqn_head: exactly one of acts (int64) / acts_f (float)");
  TRL_REQUIRE(!ring || (slots > 0 && update_count), "dqn_head: ring without slots / counter");
  TRL_REQUIRE(((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(h_next) | reinterpret_cast<uintptr_t>(w) |
                reinterpret_cast<uintptr_t>(w_t) | reinterpret_cast<uintptr_t>(dh) | reinterpret_cast<uintptr_t>(workspace)) & 15) == 0,
              "dqn_head: 16-byte aligned activations, weights, dh and workspace");
  DqHead d{};
  d.h = h; d.hn = h_next; d.w = w; d.bias = bias; d.wt = w_t; d.bias_t = bias_t; d.act = acts; d.act_f = acts_f;
  d.rew = rewards; d.term = terminals; d.gamma = gamma; d.B = B; d.H = H; d.A = A; d.dh = dh; d.dw = dw; d.db = db;
  d.q_out = q_out; d.qn_out = qn_out; d.sums = sums; d.ring = DqRing{ring, update_count, slots};
  d.part = (unsigned long long*)workspace;
  d.arrive = d.part + (size_t)DQH_WGS * (A * H + A + 6);
  const int lds = (2 * A * H + DQH_WAVES * H) * (int)sizeof(float);
  return H <= 512 ? launch_dqn_head<2>(d, lds, (hipStream_t)stream) : launch_dqn_head<4>(d, lds, (hipStream_t)stream);
}

// ---------------------------------------------------------------- K17
// action[n] = argmax_a score(n, a)  with score = Q (DQN) or mean over quantiles (QR-DQN);
// where u[n] < epsilon the action is replaced by rand_act[n]
// (discrete_policies.py:58-65: u = np.random.rand, rand_act = np.random.randint on the host in
// parity mode).  u / rand_act may be NULL (greedy).
__global__ __launch_bounds__(DQ_THREADS) void eps_greedy_kernel(const float* __restrict__ q, int N, int A, int Q,
                                                                const float* __restrict__ u, const int64_t* __restrict__ ra,
                                                                float epsilon, int64_t* __restrict__ action,
                                                                int64_t* __restrict__ ring_row, int n_rows) {
  // (a captured sequence of vector steps: the replay ring's row advances here, see dqn_act_kernel)
  if (ring_row && blockIdx.x == 0 && threadIdx.x == 0) ring_row[0] = (ring_row[0] + 1) % n_rows;
  // Q == 1: a thread per env.  Quantile nets: a wave per env, lanes stride over the quantiles of an action
  // (coalesced reads of the A * Q row; a thread per env walked it with a stride of A * Q floats).
  int best = 0;
  float bv = -INFINITY;
  int n;
  if (Q == 1) {
    n = blockIdx.x * DQ_THREADS + threadIdx.x;
    if (n >= N) return;
    for (int a = 0; a < A; ++a) {
      const float s = q[(size_t)n * A + a];
      if (s > bv) { bv = s; best = a; }
    }
  } else {
    const int lane = threadIdx.x & 63;
    n = blockIdx.x * (DQ_THREADS / 64) + (threadIdx.x >> 6);
    if (n >= N) return;                                            // wave-uniform
    for (int a = 0; a < A; ++a) {
      const float* row = q + ((size_t)n * A + a) * Q;
      float s = 0.0f;
      for (int i = lane; i < Q; i += 64) s += row[i];
      s = wave_sum(s) / (float)Q;
      if (s > bv) { bv = s; best = a; }
    }
    if (lane != 0) return;
  }
  if (u && ra && u[n] < epsilon) best = (int)ra[n];
  action[n] = best;
}
extern "C" int trl_eps_greedy_i64(const float* q, int N, int A, int Q, const float* u, const int64_t* rand_act,
                                  float epsilon, int64_t* action, int64_t* ring_row, int n_rows, void* stream) {
  TRL_REQUIRE(N >= 0 && A > 0 && Q > 0 && (!ring_row || n_rows > 0), "bad sizes");
  if (N == 0) return TRL_OK;
  TRL_REQUIRE(q && action, "null pointer");
  hipLaunchKernelGGL(eps_greedy_kernel, dim3(trl_ceil_div(N, Q == 1 ? DQ_THREADS : DQ_THREADS / 64)), dim3(DQ_THREADS), 0, (hipStream_t)stream, q,
                     N, A, Q, u, rand_act, epsilon, action, ring_row, n_rows);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// K17b: the A <= 8 wide linear head of a Q network and the epsilon-greedy action in ONE launch (nets.py:34-52's last
// nn.Linear + discrete_policies.py:40-67): q[n][a] = h[n] . w[a] + b[a], action = argmax_a q (first maximum, like torch.max),
// replaced by rand_act[n] where u[n] < epsilon.  On 512 rows the head as a GEMM was 16 workgroups + a split-reduction fold
// + the action launch: three dependent launches of 5-7 us for 6 dot products per row.  A wave owns a row; W sits in LDS.
#define DQA_MAX_A 8
__global__ __launch_bounds__(256) void dqn_act_kernel(const float* __restrict__ h, const float* __restrict__ w,
                                                      const float* __restrict__ bias, int N, int H, int A,
                                                      const float* __restrict__ u, const int64_t* __restrict__ ra, float epsilon,
                                                      float* __restrict__ q_out, int64_t* __restrict__ action,
                                                      int64_t* __restrict__ ring_row, int n_rows) {
  extern __shared__ __attribute__((aligned(16))) float ws[];          // [A][H]
  // (a captured sequence of vector steps: the replay ring's row advances here, between the frame step that filed row r
  // -- finished, stream order -- and the one that will file row r + 1; nobody of this launch reads it)
  if (ring_row && blockIdx.x == 0 && threadIdx.x == 0) ring_row[0] = (ring_row[0] + 1) % n_rows;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 4 + wave;
  const int H4 = H >> 2;
  // this row's activations are requested before the weights are staged
  f32x4 hv[4];                                                         // H <= 1024: four 16-byte pieces per lane
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k4 = lane + 64 * j;
    hv[j] = (n < N && k4 < H4) ? reinterpret_cast<const f32x4*>(h + (size_t)n * H)[k4] : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int e = threadIdx.x; e < A * H4; e += 256) reinterpret_cast<f32x4*>(ws)[e] = reinterpret_cast<const f32x4*>(w)[e];
  __syncthreads();
  if (n >= N) return;
  float part[DQA_MAX_A];
#pragma unroll
  for (int a = 0; a < DQA_MAX_A; ++a) {
    float s = 0.0f;
    if (a < A) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k4 = lane + 64 * j;
        if (k4 < H4) {
          const f32x4 wv = reinterpret_cast<const f32x4*>(ws)[a * H4 + k4];
          s = fmaf(hv[j][0], wv[0], s); s = fmaf(hv[j][1], wv[1], s); s = fmaf(hv[j][2], wv[2], s); s = fmaf(hv[j][3], wv[3], s);
        }
      }
    }
    part[a] = s;
  }
#pragma unroll
  for (int sft = 32; sft > 0; sft >>= 1) {
#pragma unroll
    for (int a = 0; a < DQA_MAX_A; ++a) part[a] += __shfl_xor(part[a], sft, 64);
  }
  if (lane == 0) {
    int best = 0;
    float bv = -INFINITY;
#pragma unroll
    for (int a = 0; a < DQA_MAX_A; ++a) {
      if (a < A) {
        const float qv = part[a] + (bias ? bias[a] : 0.0f);
        if (q_out) q_out[(size_t)n * A + a] = qv;
        if (qv > bv) { bv = qv; best = a; }
      }
    }
    if (u && ra && u[n] < epsilon) best = (int)ra[n];
    action[n] = best;
  }
}
extern "C" int trl_dqn_act_supported(int H, int A) { return H >= 4 && (H & 3) == 0 && H <= 1024 && A >= 1 && A <= DQA_MAX_A; }
extern "C" int trl_dqn_act_f32(const float* h, const float* w, const float* bias, int N, int H, int A, const float* u,
                               const int64_t* rand_act, float epsilon, float* q_out, int64_t* action, int64_t* ring_row,
                               int n_rows, void* stream) {
  TRL_REQUIRE(N >= 0 && trl_dqn_act_supported(H, A), "dqn_act: H % 4 == 0, H <= 1024, A <= 8");
  if (N == 0) return TRL_OK;
  TRL_REQUIRE(h && w && action && (!ring_row || n_rows > 0), "null pointer / ring row without a row count");
  TRL_REQUIRE(((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(w)) & 15) == 0, "dqn_act: 16-byte aligned rows");
  hipLaunchKernelGGL(dqn_act_kernel, dim3(trl_ceil_div(N, 4)), dim3(256), A * H * (int)sizeof(float), (hipStream_t)stream, h, w,
                     bias, N, H, A, u, rand_act, epsilon, q_out, action, ring_row, n_rows);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// ---------------------------------------------------------------- synthetic Atari-shaped env
// Frame stacks (N, C=4, 84, 84) uint8.  A step shifts the stack by one frame and appends a new
// pseudo-random frame: bytes [16 blk, 16 blk + 16) of the frame of env n at env-time t are the
// Philox4x32-10 block keyed (env_seed; t, 0, blk, 'FRME') (little-endian words).  A reset
// rebuilds the stack as the frames of times -(C-1) .. 0.  reward = 1 if action == (first byte of
// the new frame) % A else 0;  done = time_limit = (t >= horizon).  Pure throughput stand-in for
// ALE (SURVEY.md 8(d) cfg 5); one workgroup per env.
#define TRL_TAG_FRAME 0x46524D45u
__device__ __forceinline__ void synth_frame_fill(uint8_t* dst, int t, int64_t env_seed, int HW) {
  for (int blk = threadIdx.x; blk * 16 < HW; blk += DQ_THREADS) {
    uint32_t x[4];
    philox4x32_10((uint32_t)t, 0u, (uint32_t)blk, TRL_TAG_FRAME, (uint32_t)(env_seed & 0xFFFFFFFFll),
                  (uint32_t)((env_seed >> 32) & 0xFFFFFFFFll), x);
    *reinterpret_cast<uint4*>(dst + blk * 16) = make_uint4(x[0], x[1], x[2], x[3]);
  }
}
// FrameRing (optional, obs != NULL): the step files the WHOLE transition into row dyn[0] of the replay ring itself -- the
// pre-step stacks (obs), the stacks it produces (next_obs), the actions as floats, rewards, terminals and time limits --
// so a captured sequence of vector steps walks the ring on its own (the row lives on the device; the reset call that ends
// a step advances it) and nothing is copied afterwards.
struct FrameRing { uint8_t* obs; uint8_t* next; float* acts; float* rew; float* done; float* tl; int64_t* dyn; int n_rows;
                   // the collector's bookkeeping and the masked reset of the same vector step (cur_step != NULL): one launch
                   // instead of three (trl_collector_bookkeep_f32's arithmetic per env; an env that ends resets its own stack)
                   int32_t* cur_step; float* ep_return; int max_frames; uint8_t* mask; double* epoch_reward; int32_t* ep_count;
                   float* ep_log; int ep_cap, step; };
__global__ __launch_bounds__(DQ_THREADS) void synth_frames_kernel(uint8_t* __restrict__ frames, const int64_t* __restrict__ act,
                                                                  int32_t* __restrict__ t_env, int64_t seed_base, int horizon,
                                                                  int A, uint8_t* __restrict__ next_out,
                                                                  float* __restrict__ rew, float* __restrict__ done,
                                                                  const uint8_t* __restrict__ reset_mask, int reset_all,
                                                                  int N, int C, int HW, FrameRing r) {
  const int n = blockIdx.x;
  uint8_t* f = frames + (size_t)n * C * HW;
  const int64_t env_seed = seed_base + n;
  const bool is_reset_call = reset_all || reset_mask;
  if (is_reset_call) {
    // (the last launch of a vector step: nobody of THIS launch reads the row)
    if (r.dyn && blockIdx.x == 0 && threadIdx.x == 0) r.dyn[0] = (r.dyn[0] + 1) % r.n_rows;
    if (!reset_all && !reset_mask[n]) return;                  // block-uniform
    for (int c = 0; c < C; ++c) synth_frame_fill(f + (size_t)c * HW, c - (C - 1), env_seed, HW);
    if (threadIdx.x == 0) t_env[n] = 0;
    return;
  }
  const int t = t_env[n] + 1;
  const size_t cell = r.obs ? (size_t)r.dyn[0] * N + n : 0;           // (ring row, env): uniform per block
  if (r.obs) {
    // the pre-step stack goes to the ring while it is shifted (every frame is read once)
    uint8_t* o = r.obs + cell * C * HW;
    for (int p = threadIdx.x * 16; p < HW; p += DQ_THREADS * 16) {
      uint4 cur = *reinterpret_cast<const uint4*>(f + p);
      *reinterpret_cast<uint4*>(o + p) = cur;
      for (int c = 0; c + 1 < C; ++c) {
        cur = *reinterpret_cast<const uint4*>(f + (size_t)(c + 1) * HW + p);
        *reinterpret_cast<uint4*>(o + (size_t)(c + 1) * HW + p) = cur;
        *reinterpret_cast<uint4*>(f + (size_t)c * HW + p) = cur;
      }
    }
    next_out = r.next + (cell - n) * C * HW;                            // (indexed by n below)
  } else {
    // shift frame c <- frame c + 1 (a thread moves the same 16-byte slots of every frame, in order)
    for (int p = threadIdx.x * 16; p < HW; p += DQ_THREADS * 16)
      for (int c = 0; c + 1 < C; ++c)
        *reinterpret_cast<uint4*>(f + (size_t)c * HW + p) = *reinterpret_cast<const uint4*>(f + (size_t)(c + 1) * HW + p);
  }
  synth_frame_fill(f + (size_t)(C - 1) * HW, t, env_seed, HW);
  __syncthreads();
  if (next_out)
    for (int p = threadIdx.x * 16; p < C * HW; p += DQ_THREADS * 16)
      *reinterpret_cast<uint4*>(next_out + (size_t)n * C * HW + p) = *reinterpret_cast<const uint4*>(f + p);
  if (threadIdx.x == 0) {
    t_env[n] = t;
    const int want = f[(size_t)(C - 1) * HW] % A;
    const float rw = ((int)act[n] == want) ? 1.0f : 0.0f, dn = t >= horizon ? 1.0f : 0.0f;
    rew[n] = rw;
    done[n] = dn;
    if (r.obs) { r.acts[cell] = (float)act[n]; r.rew[cell] = rw; r.done[cell] = dn; r.tl[cell] = dn; }   // synthetic env: time_limit == done
  }
  if (r.cur_step) {
    __shared__ int s_flag;
    if (threadIdx.x == 0) {
      const float rw = rew[n];
      const bool d = done[n] != 0.0f;
      const int cs = r.cur_step[n] + 1;
      float er = r.ep_return[n] + rw;
      if (d) {
        const int slot = atomicAdd(r.ep_count, 1);
        if (slot < r.ep_cap) { r.ep_log[slot * 3 + 0] = (float)r.step; r.ep_log[slot * 3 + 1] = (float)n; r.ep_log[slot * 3 + 2] = er; }
        er = 0.0f;
      }
      const bool flag = d || cs >= r.max_frames;
      r.cur_step[n] = flag ? 0 : cs;
      r.ep_return[n] = er;
      r.mask[n] = flag ? 1 : 0;
      if (r.epoch_reward && rw != 0.0f) atomicAdd(r.epoch_reward, (double)rw);   // (a zero changes nothing; no return value awaited)
      s_flag = flag ? 1 : 0;
    }
    __syncthreads();                               // (also: the ring copy of the post-step stack has been read from f)
    if (s_flag) {
      for (int c = 0; c < C; ++c) synth_frame_fill(f + (size_t)c * HW, c - (C - 1), env_seed, HW);
      if (threadIdx.x == 0) t_env[n] = 0;
    }
  }
}
extern "C" int trl_synth_frames_step_u8(uint8_t* frames, const int64_t* acts, int32_t* t_env, int64_t env_seed_base,
                                        int horizon, int A, uint8_t* next_obs, float* rewards, float* dones, int N, int C,
                                        int HW, void* stream) {
  TRL_REQUIRE(N >= 0 && C > 0 && HW > 0 && HW % 16 == 0 && A > 0, "bad sizes (HW must be a multiple of 16)");
  if (N == 0) return TRL_OK;
  TRL_REQUIRE(frames && t_env && acts && rewards && dones, "null pointer");
  hipLaunchKernelGGL(synth_frames_kernel, dim3(N), dim3(DQ_THREADS), 0, (hipStream_t)stream, frames, acts, t_env,
                     env_seed_base, horizon, A, next_obs, rewards, dones, (const uint8_t*)nullptr, 0, N, C, HW, FrameRing{});
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}
extern "C" int trl_synth_frames_collect_u8(uint8_t* frames, const int64_t* acts, int32_t* t_env, int64_t env_seed_base,
                                           int horizon, int A, uint8_t* ring_obs, uint8_t* ring_next_obs, float* ring_acts,
                                           float* ring_rewards, float* ring_terminals, float* ring_time_limits,
                                           int64_t* ring_row, int n_rows, float* step_rewards, float* step_dones,
                                           int32_t* cur_step, float* ep_return, int max_frames, uint8_t* mask,
                                           double* epoch_reward, int32_t* ep_count, float* ep_log, int ep_cap, int step,
                                           int N, int C, int HW, void* stream) {
  TRL_REQUIRE(N >= 0 && C > 0 && HW > 0 && HW % 16 == 0 && A > 0 && n_rows > 0, "bad sizes (HW must be a multiple of 16)");
  if (N == 0) return TRL_OK;
  TRL_REQUIRE(frames && t_env && acts && step_rewards && step_dones, "null pointer");
  TRL_REQUIRE(ring_obs && ring_next_obs && ring_acts && ring_rewards && ring_terminals && ring_time_limits && ring_row,
              "null ring pointer");
  TRL_REQUIRE(!cur_step || (ep_return && mask && ep_count && ep_log && ep_cap > 0), "bookkeeping: null pointer");
  FrameRing r{ring_obs, ring_next_obs, ring_acts, ring_rewards, ring_terminals, ring_time_limits, ring_row, n_rows,
              cur_step, ep_return, max_frames, mask, epoch_reward, ep_count, ep_log, ep_cap, step};
  hipLaunchKernelGGL(synth_frames_kernel, dim3(N), dim3(DQ_THREADS), 0, (hipStream_t)stream, frames, acts, t_env,
                     env_seed_base, horizon, A, (uint8_t*)nullptr, step_rewards, step_dones, (const uint8_t*)nullptr, 0, N, C,
                     HW, r);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}
extern "C" int trl_synth_frames_reset_u8(uint8_t* frames, int32_t* t_env, int64_t env_seed_base, const uint8_t* mask,
                                         int64_t* ring_row, int n_rows, int N, int C, int HW, void* stream) {
  TRL_REQUIRE(N >= 0 && C > 0 && HW > 0 && HW % 16 == 0, "bad sizes (HW must be a multiple of 16)");
  TRL_REQUIRE(!ring_row || n_rows > 0, "ring row without a row count");
  if (N == 0) return TRL_OK;
  TRL_REQUIRE(frames && t_env, "null pointer");
  FrameRing r{};
  r.dyn = ring_row; r.n_rows = n_rows;
  hipLaunchKernelGGL(synth_frames_kernel, dim3(N), dim3(DQ_THREADS), 0, (hipStream_t)stream, frames,
                     (const int64_t*)nullptr, t_env, env_seed_base, 1, 1, (uint8_t*)nullptr, (float*)nullptr,
                     (float*)nullptr, mask, mask ? 0 : 1, N, C, HW, r);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}
