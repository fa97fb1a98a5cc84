// K8 + K9 + K10 -- fused PPO minibatch gradient on fp32 MFMA, plus the partial
// reduction, the global-norm-clip + Adam step (K11) and MLP inference.
//
// Replaces the tensor work of PPO.update (torchrl/algo/on_policy/ppo.py:41-152)
// for one minibatch.  One launch covers both networks: workgroups [0, n_wg/2)
// run the policy, [n_wg/2, n_wg) the value function; a workgroup is 4 waves
// (one per SIMD), each wave walks 32-sample tiles with a grid stride.  Per tile
// and network (D=17, H=64):
//     forward  L1 18 + L2 64 MFMA,   dH1 64,   dW2 64,   dW1 32   (v_mfma_f32_32x32x2_f32)
// = 242 MFMA * 64 cycles; heads, losses, tanh and the tile transposes run on
// the VALU / LDS next to them.  Inputs are read straight from the time-major
// rollout tensors through the minibatch's row index (the gather of
// on_policy.py:84-88 is fused away): 104 B per sample (obs 68, act 24, adv 4,
// ret 4, V_old 4) + 4 B cached old log-prob.  Weight gradients accumulate in
// MFMA accumulators across all tiles of a wave, are folded across the 4 waves
// in a fixed order (deterministic) and written as one partial per workgroup.
//
// log pi_old is read from the rollout (`old_logp`, written by the collector
// kernel with the epoch-start parameters) instead of re-running target_pf on
// every minibatch (ppo.py:54-56) -- identical values, SURVEY.md section 8(d).
#include "trl_common.h"
#include "trl_mlp.h"

#define PPO_THREADS 256
#define PPO_WAVES 4

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_sched_barrier(0);
}
// keep the scheduler from hoisting a later phase's LDS weight fetches over this point
// (it otherwise front-loads hundreds of ds_reads and spills)
#define PHASE_FENCE() __builtin_amdgcn_sched_barrier(0)

struct PpoDev {
  const float *obs, *acts, *advs, *rets, *old_values, *old_logp;
  const int64_t* row_idx;
  int rows_mb, N;
  const double* adv_raw;
  double n_global;
  const float *pf_params, *vf_params;
  float clip_para, entropy_coeff;
  int clipped_value_loss, tanh_action;
  float* partial;
  double* scal_partial;
  int n_wg, p_stride;
};

template <int D, int H, int A> struct PpoShape {
  static constexpr int NT = H / 32;
  static constexpr int KS = ksteps_for(D);
  static constexpr int XS = align4(32 * D);            // x tile, row-major [32][D]
  static constexpr int DO = 32 * 8;                    // dout stage [32][8]
  static constexpr int TS = H * TRL_TLD;               // transpose scratch
  static constexpr int WAVE_SCR = align4(TS) + XS + DO;
  static constexpr int PAR = (MlpLds<D, H, A>::SIZE > MlpLds<D, H, 1>::SIZE) ? MlpLds<D, H, A>::SIZE : MlpLds<D, H, 1>::SIZE;
  static constexpr int P_PF = MlpFlat<D, H, A>::P_PF, P_VF = MlpFlat<D, H, 1>::P_VF;
  static constexpr int P_STRIDE = ((P_PF > P_VF ? P_PF : P_VF) + 63) & ~63;
  static constexpr int SCR_ALL = PPO_WAVES * WAVE_SCR;
  static constexpr int LDS_FLOATS = align4(PAR) + (SCR_ALL > P_STRIDE ? SCR_ALL : P_STRIDE);
};

// One network (policy if O == A and IS_PF, value if O == 1) over this workgroup's tiles.
template <int D, int H, int A, int ACT, bool IS_PF>
__device__ void ppo_net_pass(const PpoDev& a, float* lds, int wg_in_net, int n_wg_net) {
  constexpr int O = IS_PF ? A : 1;
  using S = PpoShape<D, H, A>;
  using L = MlpLds<D, H, O>;
  using F = MlpFlat<D, H, O>;
  constexpr int NT = H / 32, KS = ksteps_for(D);
  constexpr int PARF = align4(S::PAR);
  constexpr int NQ = (O + 3) / 4;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, j = i, hi = lane >> 5;
  float* sp = lds;                                            // parameters
  float* scr = lds + PARF + wave * S::WAVE_SCR;               // wave-private scratch
  float* T = scr;
  float* xs = scr + align4(S::TS);
  float* douts = xs + S::XS;

  L::load(sp, IS_PF ? a.pf_params : a.vf_params, IS_PF, tid, PPO_THREADS);
  __syncthreads();

  // policy constants
  float ls[O], inv_var[O], ls_pass[O];
  if constexpr (IS_PF) {
#pragma unroll
    for (int o = 0; o < O; ++o) {
      const float raw = sp[L::LS + o];
      ls[o] = fminf(fmaxf(raw, -20.0f), 2.0f);               // continuous_policy.py:8-9,185
      ls_pass[o] = (raw >= -20.0f && raw <= 2.0f) ? 1.0f : 0.0f;
      const float sd = __expf(ls[o]);
      inv_var[o] = 1.0f / (sd * sd);
    }
  }
  // advantage normalisation constants (ppo.py:141-147): mean, unbiased std
  const double ng = a.n_global;
  const double adv_mean = a.adv_raw[0] / ng;
  const double adv_var = (a.adv_raw[1] - a.adv_raw[0] * a.adv_raw[0] / ng) / (ng - 1.0);
  const float adv_mu = (float)adv_mean;
  const float adv_rstd = 1.0f / ((float)sqrt(fmax(adv_var, 0.0)) + 1e-5f);
  const float inv_b = (float)(1.0 / ng);

  // gradient accumulators (lane = input feature i, reg r = output feature rowmap(r,hi) of tile ma)
  f32x16 gW2[NT][NT], gW1[NT];
  float gW3[O][NT], gb1[NT], gb2[NT], gb3[O], gls[O];
#pragma unroll
  for (int ma = 0; ma < NT; ++ma) {
    gW1[ma] = zero_tile(); gb1[ma] = 0.f; gb2[ma] = 0.f;
#pragma unroll
    for (int mb = 0; mb < NT; ++mb) gW2[ma][mb] = zero_tile();
  }
#pragma unroll
  for (int o = 0; o < O; ++o) {
    gb3[o] = 0.f; gls[o] = 0.f;
#pragma unroll
    for (int m = 0; m < NT; ++m) gW3[o][m] = 0.f;
  }
  // scalar statistics (lanes hi == 0 only)
  double st_sum = 0.0, st_sq = 0.0, st_loss = 0.0;
  float st_max = -INFINITY, st_nmin = -INFINITY, st_rmax = -INFINITY, st_nrmin = -INFINITY;

  const int B = a.rows_mb * a.N;
  const int n_tiles = (B + 31) / 32;
  const bool contig = (a.N % 32) == 0;

  for (int tile = wg_in_net * PPO_WAVES + wave; tile < n_tiles; tile += n_wg_net * PPO_WAVES) {
    const int s0 = tile * 32;
    const int s = s0 + j;
    const bool valid = s < B;
    // flat (row, env) position of this lane's sample in the (rows, N, feat) tensors
    int64_t pos = 0;
    if (valid) {
      const int r = s / a.N, e = s - r * a.N;
      pos = (a.row_idx ? a.row_idx[r] : (int64_t)r) * a.N + e;
    }
    // ---- stage the x tile (row-major [32][D]) ----
    if (contig) {
      const int r0 = s0 / a.N, e0 = s0 - r0 * a.N;
      const float* src = a.obs + ((a.row_idx ? a.row_idx[r0] : (int64_t)r0) * a.N + e0) * D;
      for (int e = lane; e < 32 * D; e += 64) xs[e] = src[e];
    } else {
      for (int e = lane; e < 32 * D; e += 64) {
        const int sj = e / D, k = e - sj * D;
        const int64_t p = __shfl(pos, sj, 64);
        xs[e] = (s0 + sj < B) ? a.obs[p * D + k] : 0.0f;
      }
    }
    wave_lds_sync();
    float xb[KS];
#pragma unroll
    for (int q = 0; q < KS; ++q) { const int k = rowmap(q, hi); xb[q] = (k < D) ? xs[j * D + k] : 0.0f; }

    // ---- forward ----
    f32x16 h1[NT], h2[NT];
#pragma unroll
    for (int mo = 0; mo < NT; ++mo)
      h1[mo] = act_tile<ACT>(layer1_tile<D, L::LD1, KS>(bias_tile(sp + L::B1 + 32 * mo, hi), sp + L::W1, mo, xb, i, hi));
    PHASE_FENCE();
#pragma unroll
    for (int mo = 0; mo < NT; ++mo) {
      h2[mo] = act_tile<ACT>(layer_tile<NT, L::LD2>(bias_tile(sp + L::B2 + 32 * mo, hi), sp + L::W2, mo, h1, i, hi));
      PHASE_FENCE();
    }
    float out[O];
    head_fwd<NT, H, O>(sp + L::W3, sp + L::B3, h2, hi, out);
    PHASE_FENCE();

    // ---- loss and d(loss)/d(out) ----
    float dout[O];
    if constexpr (IS_PF) {
      const float advn = valid ? (a.advs[pos] - adv_mu) * adv_rstd : 0.0f;
      const float lp_old = valid ? a.old_logp[pos] : 0.0f;
      float zc[O];
      float lp = 0.0f;
#pragma unroll
      for (int o = 0; o < O; ++o) {
        const float act = valid ? a.acts[pos * O + o] : 0.0f;
        float pre = act, corr = 0.0f;
        if (a.tanh_action) {                                   // distribution.py:40-45
          pre = 0.5f * logf((1.0f + act) / (1.0f - act));
          corr = logf(1.0f - act * act + 1e-6f);
        }
        zc[o] = pre - out[o];
        lp += -(zc[o] * zc[o]) * 0.5f * inv_var[o] - ls[o] - 0.91893853320467274f - corr;
      }
      const float ratio = __expf(lp - lp_old);
      const float s1 = ratio * advn;
      const float s2 = fminf(fmaxf(ratio, 1.0f - a.clip_para), 1.0f + a.clip_para) * advn;
      const float g_lp = (valid && s1 <= s2) ? -advn * ratio * inv_b : 0.0f;
#pragma unroll
      for (int o = 0; o < O; ++o) {
        dout[o] = g_lp * zc[o] * inv_var[o];
        if (hi == 0 && valid)
          gls[o] += ls_pass[o] * (g_lp * (zc[o] * zc[o] * inv_var[o] - 1.0f) - a.entropy_coeff * inv_b);
      }
      if (hi == 0 && valid) {
        st_sum += (double)lp; st_sq += (double)lp * (double)lp; st_loss += (double)(-fminf(s1, s2));
        st_max = fmaxf(st_max, lp); st_nmin = fmaxf(st_nmin, -lp);
        st_rmax = fmaxf(st_rmax, ratio); st_nrmin = fmaxf(st_nrmin, -ratio);
      }
    } else {
      const float v = out[0];
      const float R = valid ? a.rets[pos] : 0.0f;
      float dv, l;
      if (a.clipped_value_loss) {                              // ppo.py:104-111
        const float vo = valid ? a.old_values[pos] : 0.0f;
        const float dc = v - vo;
        const float vc = vo + fminf(fmaxf(dc, -a.clip_para), a.clip_para);
        const float l1 = (v - R) * (v - R), l2 = (vc - R) * (vc - R);
        const float w1 = l1 > l2 ? 1.0f : (l1 == l2 ? 0.5f : 0.0f), w2 = 1.0f - w1;
        const float pass = (dc >= -a.clip_para && dc <= a.clip_para) ? 1.0f : 0.0f;
        l = 0.5f * fmaxf(l1, l2);
        dv = inv_b * (w1 * (v - R) + w2 * pass * (vc - R));
      } else {                                                 // nn.MSELoss, a2c.py:43
        l = (v - R) * (v - R);
        dv = 2.0f * (v - R) * inv_b;
      }
      dout[0] = valid ? dv : 0.0f;
      if (hi == 0 && valid) st_loss += (double)l;
    }
#pragma unroll
    for (int o = 0; o < O; ++o) if (hi == 0) gb3[o] += dout[o];

    // ---- backward through the head ----
    PHASE_FENCE();
    f32x16 dz2[NT];
    head_bwd<NT, H, O>(sp + L::W3, dout, hi, dz2);
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) dz2[m][r] *= act_grad<ACT>(h2[m][r]);

    // dW3[o][f] += sum_j dout[o][j] H2[j][f]  -- needs H2 with lane = feature
    tile_store_T<NT>(T, h2, j, hi);
    if (hi == 0) {
#pragma unroll
      for (int o = 0; o < 4 * NQ; ++o) douts[j * 8 + o] = (o < O) ? dout[o] : 0.0f;
    }
    wave_lds_sync();
#pragma unroll
    for (int m = 0; m < NT; ++m) {
      const f32x16 h2n = tile_load_N(T, m, i, hi);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const f32x4 d = *reinterpret_cast<const f32x4*>(douts + rowmap(r, hi) * 8 + 4 * q);
#pragma unroll
          for (int c = 0; c < 4; ++c) if (4 * q + c < O) gW3[4 * q + c][m] = fmaf(h2n[r], d[c], gW3[4 * q + c][m]);
        }
      }
    }
    wave_lds_sync();

    // ---- dH1^T = W2^T dZ2^T, dZ1 = dH1 * act'(H1) ----
    PHASE_FENCE();
    f32x16 dz1[NT];
#pragma unroll
    for (int mo = 0; mo < NT; ++mo) {
      dz1[mo] = layer_tile_wT<NT, L::LD2>(zero_tile(), sp + L::W2, mo, dz2, i, hi);
#pragma unroll
      for (int r = 0; r < 16; ++r) dz1[mo][r] *= act_grad<ACT>(h1[mo][r]);
    }

    // ---- dW2^T[j_out][k_in] += sum_s dZ2[s][j_out] H1[s][k_in] ----
    PHASE_FENCE();
    f32x16 h1n[NT];
    tile_store_T<NT>(T, h1, j, hi);
    wave_lds_sync();
#pragma unroll
    for (int m = 0; m < NT; ++m) h1n[m] = tile_load_N(T, m, i, hi);
    wave_lds_sync();
    tile_store_T<NT>(T, dz2, j, hi);
    wave_lds_sync();
#pragma unroll
    for (int ma = 0; ma < NT; ++ma) {
      const f32x16 dzn = tile_load_N(T, ma, i, hi);
      float bs = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) bs += dzn[r];
      gb2[ma] += bs;
#pragma unroll
      for (int mb = 0; mb < NT; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) gW2[ma][mb] = mfma32(dzn[r], h1n[mb][r], gW2[ma][mb]);
    }
    wave_lds_sync();

    // ---- dW1^T[j_out][k_in] += sum_s dZ1[s][j_out] X[s][k_in] ----
    tile_store_T<NT>(T, dz1, j, hi);
    wave_lds_sync();
    float xn[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) xn[r] = (i < D) ? xs[rowmap(r, hi) * D + i] : 0.0f;
#pragma unroll
    for (int ma = 0; ma < NT; ++ma) {
      const f32x16 dzn = tile_load_N(T, ma, i, hi);
      float bs = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) bs += dzn[r];
      gb1[ma] += bs;
#pragma unroll
      for (int r = 0; r < 16; ++r) gW1[ma] = mfma32(dzn[r], xn[r], gW1[ma]);
    }
    wave_lds_sync();
  }

  // ---- fold the 4 waves in fixed order into one partial gradient (flat layout) ----
  __syncthreads();
  float* gacc = lds + PARF;                                   // reuse scratch: S::P_STRIDE floats
  for (int e = tid; e < S::P_STRIDE; e += PPO_THREADS) gacc[e] = 0.0f;
  __syncthreads();
  for (int w = 0; w < PPO_WAVES; ++w) {
    if (wave == w) {
#pragma unroll
      for (int ma = 0; ma < NT; ++ma) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int jo = 32 * ma + rowmap(r, hi);
          if (i < D) gacc[F::W1 + jo * D + i] += gW1[ma][r];
#pragma unroll
          for (int mb = 0; mb < NT; ++mb) gacc[F::W2 + jo * H + 32 * mb + i] += gW2[ma][mb][r];
        }
        // per-feature partials live in both hi halves: fold with one shuffle
        const float b1v = gb1[ma] + __shfl_xor(gb1[ma], 32, 64);
        const float b2v = gb2[ma] + __shfl_xor(gb2[ma], 32, 64);
        if (hi == 0) { gacc[F::B1 + 32 * ma + i] += b1v; gacc[F::B2 + 32 * ma + i] += b2v; }
#pragma unroll
        for (int o = 0; o < O; ++o) {
          const float w3v = gW3[o][ma] + __shfl_xor(gW3[o][ma], 32, 64);
          if (hi == 0) gacc[F::W3 + o * H + 32 * ma + i] += w3v;
        }
      }
#pragma unroll
      for (int o = 0; o < O; ++o) {
        const float b3v = wave_sum(gb3[o]);                   // hi == 1 lanes hold 0
        if (lane == 0) gacc[F::B3 + o] += b3v;
        if (IS_PF) { const float lv = wave_sum(gls[o]); if (lane == 0) gacc[F::LS + o] += lv; }
      }
    }
    __syncthreads();
  }
  const int wg = blockIdx.x;
  for (int e = tid; e < S::P_STRIDE; e += PPO_THREADS) a.partial[(size_t)wg * a.p_stride + e] = gacc[e];

  // ---- scalar statistics: wave shuffle reduce, then across waves through LDS ----
  __syncthreads();
  double* sred = reinterpret_cast<double*>(lds + PARF);
  {
    const double v0 = wave_sum(st_sum), v1 = wave_sum(st_sq), v6 = wave_sum(st_loss);
    const float v2 = wave_max(st_max), v3 = wave_max(st_nmin), v4 = wave_max(st_rmax), v5 = wave_max(st_nrmin);
    if (lane == 0) {
      double* p = sred + wave * 8;
      p[0] = v0; p[1] = v1; p[2] = v2; p[3] = v3; p[4] = v4; p[5] = v5; p[6] = v6; p[7] = 0.0;
    }
  }
  __syncthreads();
  if (tid < 8) {
    double r = sred[tid];
    for (int w = 1; w < PPO_WAVES; ++w) {
      const double o = sred[w * 8 + tid];
      r = (tid >= 2 && tid <= 5) ? fmax(r, o) : r + o;
    }
    a.scal_partial[(size_t)wg * 8 + tid] = r;
  }
}

template <int D, int H, int A, int ACT>
__global__ __launch_bounds__(PPO_THREADS, 1) void ppo_grad_kernel(PpoDev a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int half = a.n_wg >> 1;
  if ((int)blockIdx.x < half) ppo_net_pass<D, H, A, ACT, true>(a, lds, blockIdx.x, half);
  else                        ppo_net_pass<D, H, A, ACT, false>(a, lds, blockIdx.x - half, half);
}

// ---------------------------------------------------------------- partial reduce
// grads[p] = sum_w partial[w][p] in fixed order; info[] from the scalar partials:
//  0 policy surrogate sum (-min(s1,s2))   1 sum logp   2 sum logp^2   3 max logp   4 -min logp
//  5 max ratio   6 -min ratio   7 value-loss sum
__global__ __launch_bounds__(256) void ppo_reduce_kernel(const float* __restrict__ partial,
                                                         const double* __restrict__ scal, int n_wg,
                                                         int p_stride, int p_pf, int p_vf,
                                                         const float* __restrict__ logstd, int n_act,
                                                         float* __restrict__ grads, double* __restrict__ info) {
  const int half = n_wg >> 1;
  const int net = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int pn = net == 0 ? p_pf : p_vf;
  if (p < pn) {
    const float* src = partial + (size_t)(net * half) * p_stride + p;
    float acc = 0.0f;
    for (int w = 0; w < half; ++w) acc += src[(size_t)w * p_stride];
    grads[(net == 0 ? 0 : p_pf) + p] = acc;
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 8) {
    const int k = threadIdx.x;
    if (k < 7) {
      double r = scal[k];
      for (int w = 1; w < half; ++w) {
        const double o = scal[(size_t)w * 8 + k];
        r = (k >= 2 && k <= 5) ? fmax(r, o) : r + o;
      }
      const int slot = (k == 6) ? 0 : k + 1;               // -> layout documented above
      info[slot] = r;
    } else {
      double r = 0.0;
      for (int w = 0; w < half; ++w) r += scal[(size_t)(half + w) * 8 + 6];
      info[7] = r;
      if (logstd) {                                          // log_std/{mean,std,max,min} (ppo.py:82-85)
        double sm = 0, sq = 0, mx = -INFINITY, mn = INFINITY;
        for (int o = 0; o < n_act; ++o) {
          const double v = fmin(fmax((double)logstd[o], -20.0), 2.0);
          sm += v; sq += v * v; mx = fmax(mx, v); mn = fmin(mn, v);
        }
        const double mean = sm / n_act;
        info[8] = mean;
        info[9] = n_act > 1 ? sqrt(fmax((sq - sm * mean) / (n_act - 1), 0.0)) : NAN;
        info[10] = mx; info[11] = mn;
      }
    }
  }
}

// ---------------------------------------------------------------- K11 clip + Adam
// Every block recomputes the (tiny) group norms itself, so one launch does
// clip_grad_norm_ (coef = max_norm / (norm + 1e-6), clamped to 1) and Adam.
struct AdamDev {
  float* params; const float* grads; float* m; float* v;
  int n_groups; int off[5]; float lr[4];
  float max_norm, beta1, beta2, eps, grad_scale, bc1, bc2_sqrt;
  float* norms_out;
};
__global__ __launch_bounds__(256) void clip_adam_kernel(AdamDev a) {
  __shared__ float s_part[4][4];
  __shared__ float s_coef[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int g = 0; g < a.n_groups; ++g) {
    float ss = 0.0f;
    for (int e = a.off[g] + tid; e < a.off[g + 1]; e += 256) { const float x = a.grads[e] * a.grad_scale; ss = fmaf(x, x, ss); }
    ss = wave_sum(ss);
    if (lane == 0) s_part[g][wave] = ss;
  }
  __syncthreads();
  if (tid < a.n_groups) {
    const float norm = sqrtf(s_part[tid][0] + s_part[tid][1] + s_part[tid][2] + s_part[tid][3]);
    s_coef[tid] = (a.max_norm > 0.0f) ? fminf(a.max_norm / (norm + 1e-6f), 1.0f) : 1.0f;
    if (blockIdx.x == 0 && a.norms_out) a.norms_out[tid] = norm;
  }
  __syncthreads();
  const int e = blockIdx.x * 256 + tid;
  if (e >= a.off[a.n_groups]) return;
  int g = 0;
  while (e >= a.off[g + 1]) ++g;
  const float gr = a.grads[e] * a.grad_scale * s_coef[g];
  const float m = a.beta1 * a.m[e] + (1.0f - a.beta1) * gr;
  const float v = a.beta2 * a.v[e] + (1.0f - a.beta2) * gr * gr;
  a.m[e] = m; a.v[e] = v;
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  a.params[e] -= (a.lr[g] / a.bc1) * (m / denom);
}

// ---------------------------------------------------------------- MLP inference
template <int D, int H, int O, int ACT>
__global__ __launch_bounds__(256) void mlp2_forward_kernel(const float* __restrict__ params,
                                                           const float* __restrict__ x,
                                                           float* __restrict__ out, int M) {
  using L = MlpLds<D, H, O>;
  constexpr int NT = H / 32, KS = ksteps_for(D);
  __shared__ __attribute__((aligned(16))) float sp[L::SIZE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, hi = lane >> 5;
  L::load(sp, params, false, tid, 256);
  __syncthreads();
  const int n_tiles = (M + 31) / 32;
  for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
    const int s = tile * 32 + i;
    float xb[KS];
#pragma unroll
    for (int q = 0; q < KS; ++q) { const int k = rowmap(q, hi); xb[q] = (k < D && s < M) ? x[(size_t)s * D + k] : 0.0f; }
    f32x16 h1[NT], h2[NT];
#pragma unroll
    for (int mo = 0; mo < NT; ++mo)
      h1[mo] = act_tile<ACT>(layer1_tile<D, L::LD1, KS>(bias_tile(sp + L::B1 + 32 * mo, hi), sp + L::W1, mo, xb, i, hi));
#pragma unroll
    for (int mo = 0; mo < NT; ++mo)
      h2[mo] = act_tile<ACT>(layer_tile<NT, L::LD2>(bias_tile(sp + L::B2 + 32 * mo, hi), sp + L::W2, mo, h1, i, hi));
    float o[O];
    head_fwd<NT, H, O>(sp + L::W3, sp + L::B3, h2, hi, o);
    if (hi == 0 && s < M) {
#pragma unroll
      for (int c = 0; c < O; ++c) out[(size_t)s * O + c] = o[c];
    }
  }
}

// ================================================================ host side
#define SHAPE_IS(d, h, o) (D == (d) && H == (h) && A == (o))

extern "C" int trl_ppo_partial_stride(int D, int H, int A) {
  if (SHAPE_IS(17, 64, 6)) return PpoShape<17, 64, 6>::P_STRIDE;
  trl_set_error("trl_ppo_partial_stride: shape D=%d H=%d A=%d not instantiated", D, H, A);
  return TRL_EUNSUPPORTED;
}

template <int D, int H, int A, int ACT>
static int launch_ppo(const PpoDev& d, hipStream_t s) {
  using S = PpoShape<D, H, A>;
  const size_t lds = S::LDS_FLOATS * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)ppo_grad_kernel<D, H, A, ACT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { trl_set_error("ppo_grad: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  hipLaunchKernelGGL((ppo_grad_kernel<D, H, A, ACT>), dim3(d.n_wg), dim3(PPO_THREADS), lds, s, d);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

extern "C" int trl_ppo_minibatch_grad_f32(const trl_ppo_batch_t* p, void* stream) {
  if (!p) { trl_set_error("ppo_grad: null descriptor"); return TRL_EINVAL; }
  TRL_REQUIRE(p->obs && p->acts && p->advs && p->rets && p->old_values && p->old_logp, "null rollout tensor");
  TRL_REQUIRE(p->adv_raw && p->pf_params && p->vf_params && p->partial && p->scal_partial, "null pointer");
  TRL_REQUIRE(p->rows_mb > 0 && p->N > 0, "empty minibatch");
  TRL_REQUIRE(p->n_wg >= 2 && (p->n_wg % 2) == 0, "n_wg must be even and >= 2");
  TRL_REQUIRE(p->n_global > 1.0, "n_global must exceed 1 (unbiased std)");
  const int D = p->D, H = p->H, A = p->A;
  PpoDev d;
  d.obs = p->obs; d.acts = p->acts; d.advs = p->advs; d.rets = p->rets; d.old_values = p->old_values;
  d.old_logp = p->old_logp; d.row_idx = p->row_idx; d.rows_mb = p->rows_mb; d.N = p->N;
  d.adv_raw = p->adv_raw; d.n_global = p->n_global; d.pf_params = p->pf_params; d.vf_params = p->vf_params;
  d.clip_para = p->clip_para; d.entropy_coeff = p->entropy_coeff;
  d.clipped_value_loss = p->clipped_value_loss; d.tanh_action = p->tanh_action;
  d.partial = p->partial; d.scal_partial = p->scal_partial; d.n_wg = p->n_wg;
  hipStream_t s = (hipStream_t)stream;
  if (SHAPE_IS(17, 64, 6)) {
    d.p_stride = PpoShape<17, 64, 6>::P_STRIDE;
    if (p->act == TRL_ACT_TANH) return launch_ppo<17, 64, 6, TRL_ACT_TANH>(d, s);
    if (p->act == TRL_ACT_RELU) return launch_ppo<17, 64, 6, TRL_ACT_RELU>(d, s);
  }
  trl_set_error("ppo_grad: shape D=%d H=%d A=%d act=%d not instantiated", D, H, A, p->act);
  return TRL_EUNSUPPORTED;
}

extern "C" int trl_ppo_reduce_f32(const float* partial, const double* scal_partial, int n_wg, int D, int H,
                                  int A, const float* pf_params, float* grads, double* info, void* stream) {
  TRL_REQUIRE(partial && scal_partial && grads && info, "null pointer");
  TRL_REQUIRE(n_wg >= 2 && (n_wg % 2) == 0, "n_wg must be even and >= 2");
  const int ps = trl_ppo_partial_stride(D, H, A);
  if (ps < 0) return ps;
  const int p_pf = H * D + H + H * H + H + A * H + A + A, p_vf = H * D + H + H * H + H + H + 1;
  hipLaunchKernelGGL(ppo_reduce_kernel, dim3(trl_ceil_div(ps, 256), 2), dim3(256), 0, (hipStream_t)stream,
                     partial, scal_partial, n_wg, ps, p_pf, p_vf,
                     pf_params ? pf_params + (p_pf - A) : (const float*)nullptr, A, grads, info);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

extern "C" int trl_clip_adam_f32(const trl_adam_t* p, void* stream) {
  if (!p) { trl_set_error("clip_adam: null descriptor"); return TRL_EINVAL; }
  TRL_REQUIRE(p->params && p->grads && p->exp_avg && p->exp_avg_sq, "null pointer");
  TRL_REQUIRE(p->n_groups >= 1 && p->n_groups <= 4, "n_groups must be 1..4");
  TRL_REQUIRE(p->step_count >= 1, "step_count starts at 1");
  AdamDev d;
  d.params = p->params; d.grads = p->grads; d.m = p->exp_avg; d.v = p->exp_avg_sq;
  d.n_groups = p->n_groups; d.off[0] = 0;
  for (int g = 0; g < p->n_groups; ++g) {
    TRL_REQUIRE(p->group_sizes[g] >= 0, "negative group size");
    d.off[g + 1] = d.off[g] + p->group_sizes[g];
    d.lr[g] = p->group_lr[g];
  }
  for (int g = p->n_groups; g < 4; ++g) { d.off[g + 1] = d.off[g]; if (g < 4) d.lr[g] = 0.f; }
  d.max_norm = p->max_norm; d.beta1 = p->beta1; d.beta2 = p->beta2; d.eps = p->eps;
  d.grad_scale = p->grad_scale;
  d.bc1 = (float)(1.0 - pow((double)p->beta1, (double)p->step_count));
  d.bc2_sqrt = (float)sqrt(1.0 - pow((double)p->beta2, (double)p->step_count));
  d.norms_out = p->norms_out;
  const int total = d.off[p->n_groups];
  if (total == 0) return TRL_OK;
  hipLaunchKernelGGL(clip_adam_kernel, dim3(trl_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, d);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

template <int D, int H, int O>
static int launch_fwd(const float* params, const float* x, float* out, int M, int act, hipStream_t s) {
  int grid = trl_ceil_div(M, 128);
  if (grid > 1024) grid = 1024;
  if (act == TRL_ACT_TANH)
    hipLaunchKernelGGL((mlp2_forward_kernel<D, H, O, TRL_ACT_TANH>), dim3(grid), dim3(256), 0, s, params, x, out, M);
  else
    hipLaunchKernelGGL((mlp2_forward_kernel<D, H, O, TRL_ACT_RELU>), dim3(grid), dim3(256), 0, s, params, x, out, M);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

extern "C" int trl_mlp2_forward_f32(const float* params, const float* x, float* out, int M, int D, int H,
                                    int O, int act, void* stream) {
  TRL_REQUIRE(M >= 0, "negative M");
  if (M == 0) return TRL_OK;
  TRL_REQUIRE(params && x && out, "null pointer");
  TRL_REQUIRE(act == TRL_ACT_TANH || act == TRL_ACT_RELU, "unknown activation");
  hipStream_t s = (hipStream_t)stream;
  if (D == 17 && H == 64 && O == 6) return launch_fwd<17, 64, 6>(params, x, out, M, act, s);
  if (D == 17 && H == 64 && O == 1) return launch_fwd<17, 64, 1>(params, x, out, M, act, s);
  trl_set_error("mlp2_forward: shape D=%d H=%d O=%d not instantiated", D, H, O);
  return TRL_EUNSUPPORTED;
}
