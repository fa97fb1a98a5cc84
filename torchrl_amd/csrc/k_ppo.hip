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

#define PPO_THREADS 512
#define PPO_WAVES 8
#define PPO_PAIRS 4

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Two waves of a pair rendezvous on an LDS counter (the other pairs of the workgroup keep
// running: a whole-workgroup s_barrier would put all 8 waves in lockstep and the MFMA phases of
// one pair could no longer overlap the VALU/LDS phases of the pair sharing its SIMDs).
__device__ __forceinline__ void pair_sync(int* cnt, int& expect, int lane) {
#ifdef TRL_EXP_NOSYNC
  return;
#endif
  // LDS operations of one wave are performed in order, so the arrive-atomic below is ordered after
  // every earlier ds_write of this wave without any s_waitcnt.  No fence intrinsic here on purpose:
  // a workgroup-scope release also emits vmcnt(0) and would stall on the global loads this kernel
  // deliberately keeps in flight across the rendezvous.
  asm volatile("" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  expect += 2;
  while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < expect) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}

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
  static_assert(H == 64, "pair-cooperative kernel: two 32-feature tiles, one per wave of a pair");
  static constexpr int KS = ksteps_for(D);
  static constexpr int TS = H * TRL_TLD;               // (64 x 33) staging tensor shared by the pair
  static constexpr int PT = 32 * TRL_TLD;              // wave-private (32 x 33) tile
  static constexpr int NSTAT = 7 + 2 * A;              // lp sum/sumsq/max/-min, ratio max/-min, loss, db3[A], dlogstd[A]
  // pair scratch: H1s | DZ2s | P[2] | douts[32][8] | headp[2][8][32] | stats[NSTAT][32] | counter
  static constexpr int O_H1 = 0, O_DZ2 = TS, O_P = 2 * TS, O_DO = O_P + 2 * PT,
                       O_HP = O_DO + 256, O_ST = O_HP + 512, O_CNT = O_ST + align4(NSTAT * 32),
                       PAIR_SCR = O_CNT + 4;
  static constexpr int PAR = (MlpLds<D, H, A>::SIZE > MlpLds<D, H, 1>::SIZE) ? MlpLds<D, H, A>::SIZE : MlpLds<D, H, 1>::SIZE;
  static constexpr int P_PF = MlpFlat<D, H, A>::P_PF, P_VF = MlpFlat<D, H, 1>::P_VF;
  static constexpr int P_STRIDE = ((P_PF > P_VF ? P_PF : P_VF) + 63) & ~63;
  static constexpr int SCR_ALL = PPO_PAIRS * PAIR_SCR;
  static constexpr int LDS_FLOATS = align4(PAR) + (SCR_ALL > P_STRIDE ? SCR_ALL : P_STRIDE);
};

// One network (policy or value) over this workgroup's tiles.  A pair of waves owns a 32-sample
// tile; wave `mo` of the pair computes the 32 hidden features [32mo, 32mo+32) of every layer,
// its half of the weight gradients, and exchanges activations with its partner through LDS.
template <int D, int H, int A, int ACT, bool IS_PF>
__device__ void ppo_net_pass(const PpoDev& a, float* lds, int wg_in_net, int n_wg_net) {
  constexpr int O = IS_PF ? A : 1;
  using S = PpoShape<D, H, A>;
  using L = MlpLds<D, H, O>;
  using F = MlpFlat<D, H, O>;
  constexpr int KS = ksteps_for(D);
  constexpr int PARF = align4(S::PAR);
  constexpr int NQ = (O + 3) / 4;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pair = wave >> 1, mo0 = wave & 1;
  const int i0 = lane & 31, hi0 = lane >> 5;
  float* sp = lds;                                            // parameters (shared by the 4 pairs)
  int* cnt = reinterpret_cast<int*>(lds + PARF + pair * S::PAIR_SCR + S::O_CNT);

  L::load(sp, IS_PF ? a.pf_params : a.vf_params, IS_PF, tid, PPO_THREADS);
  if (lane == 0 && mo0 == 0) *cnt = 0;
  {
    float* st0 = lds + PARF + pair * S::PAIR_SCR + S::O_ST;  // per-lane statistic slots, [k][32]
    if (mo0 == 0 && hi0 == 0)
      for (int k = 0; k < S::NSTAT; ++k) st0[k * 32 + i0] = (k >= 2 && k <= 5) ? -INFINITY : 0.0f;
  }
  __syncthreads();
  int expect = 0;

  // advantage normalisation constants (ppo.py:141-147): mean, unbiased std
  const double ng = a.n_global;
  const double adv_mean = a.adv_raw[0] / ng;
  const double adv_var = (a.adv_raw[1] - a.adv_raw[0] * a.adv_raw[0] / ng) / (ng - 1.0);
  const float adv_mu = (float)adv_mean;
  const float adv_rstd = 1.0f / ((float)sqrt(fmax(adv_var, 0.0)) + 1e-5f);
  const float inv_b = (float)(1.0 / ng);

  // this wave's share of the gradient: rows (output features) [32mo, 32mo+32) of W2^T / W1^T
  f32x16 gW2[2], gW1;
  float gW3[O], gb1 = 0.f, gb2 = 0.f;
  gW2[0] = zero_tile(); gW2[1] = zero_tile(); gW1 = zero_tile();
#pragma unroll
  for (int o = 0; o < O; ++o) gW3[o] = 0.f;
  // scalar statistics, db3 and dlogstd are touched once per tile by 32 lanes only: they live in
  // per-lane LDS slots (wave mo == 0, lanes hi == 0) instead of registers

  const int B = a.rows_mb * a.N;
  const int n_tiles = (B + 31) / 32;
  const bool contig = (a.N % 32) == 0;
  const bool stat_lane = (mo0 == 0 && hi0 == 0);

  int64_t pos_next = 0;
  {
    const int sf = (wg_in_net * PPO_PAIRS + pair) * 32 + i0;
    if (sf < B) {
      const int r = sf / a.N, e = sf - r * a.N;
      pos_next = (a.row_idx ? a.row_idx[r] : (int64_t)r) * a.N + e;
    }
  }
  for (int tile = wg_in_net * PPO_PAIRS + pair; tile < n_tiles; tile += n_wg_net * PPO_PAIRS) {
    // Launder the lane coordinates once per tile: every LDS address below derives from them, and
    // without this LICM hoists ~100 loop-invariant addresses out of the tile loop and spills them.
    int i = i0, hi = hi0, mo = mo0, scr_off = PARF + pair * S::PAIR_SCR;
    asm volatile("" : "+v"(i), "+v"(hi), "+v"(mo), "+v"(scr_off));
    const int j = i, mx = mo ^ 1;
    float* scr = lds + scr_off;
    float* H1s = scr + S::O_H1;
    float* DZ2s = scr + S::O_DZ2;
    float* P = scr + S::O_P + mo * S::PT;                     // wave private
    float* douts = scr + S::O_DO;
    float* headp = scr + S::O_HP;
    const int s0 = tile * 32;
    const int s = s0 + j;
    const bool valid = s < B;
    const int64_t pos = pos_next;                             // (row, env) cell of this lane's sample (0 if masked)
    {                                                         // next tile's cell: its row_idx load flies during this tile
      const int sn = s + n_wg_net * PPO_PAIRS * 32;
      int64_t pn = 0;
      if (sn < B) {
        const int r = sn / a.N, e = sn - r * a.N;
        pn = (a.row_idx ? a.row_idx[r] : (int64_t)r) * a.N + e;
      }
      pos_next = pn;
    }
    // this tile's per-sample scalars, issued now and consumed after layer 2 (latency hidden by the MFMAs)
    float in_act[IS_PF ? O : 1], in_a, in_b;
    if constexpr (IS_PF) {
#pragma unroll
      for (int o = 0; o < O; ++o) in_act[o] = a.acts[pos * O + o];
      in_a = a.advs[pos]; in_b = a.old_logp[pos];
    } else {
      in_act[0] = 0.0f;
      in_a = a.rets[pos]; in_b = a.clipped_value_loss ? a.old_values[pos] : 0.0f;
    }
    float* st = scr + S::O_ST;
    // ---- x^T operand straight from HBM/L2: lane (sample j, hi) holds features rowmap(q, hi) ----
    float xb[KS];
#pragma unroll
    for (int q = 0; q < KS; ++q) {                            // unconditional (clamped) loads + select: no exec-mask branches
      const int k = rowmap(q, hi);
      const float v = a.obs[pos * D + (k < D ? k : D - 1)];
      xb[q] = (valid && k < D) ? v : 0.0f;
    }

    // ---- forward layer 1, own feature tile ----
    const f32x16 h1 = act_tile<ACT>(layer1_tile<D, L::LD1, KS>(bias_tile(sp + L::B1 + 32 * mo, hi), sp + L::W1, mo, xb, i, hi));
    pair_sync(cnt, expect, lane);                             // partner is done with the previous tile's H1s / DZ2s
    tile_store_T1(H1s, mo, h1, j, hi);
    pair_sync(cnt, expect, lane);

    // ---- forward layer 2: own half from registers, partner's half from LDS ----
    f32x16 h2;
    {
      const f32x16 h1x = tile_load_T1(H1s, mx, j, hi);
      f32x16 acc = bias_tile(sp + L::B2 + 32 * mo, hi);
      acc = layer_tile_1src<L::LD2>(acc, sp + L::W2, mo, mo, h1, i, hi);
      acc = layer_tile_1src<L::LD2>(acc, sp + L::W2, mo, mx, h1x, i, hi);
      h2 = act_tile<ACT>(acc);
    }
    // partial head over the own 32 features
#pragma unroll
    for (int o = 0; o < O; ++o) {
      float p = 0.0f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(sp + L::W3 + o * H + 32 * mo + 8 * q + 4 * hi);
        p = fmaf(w[0], h2[4 * q + 0], p); p = fmaf(w[1], h2[4 * q + 1], p);
        p = fmaf(w[2], h2[4 * q + 2], p); p = fmaf(w[3], h2[4 * q + 3], p);
      }
      p += __shfl_xor(p, 32, 64);
      if (hi == 0) headp[(mo * 8 + o) * 32 + j] = p;
    }
    tile_store_T1(P, 0, h2, j, hi);                           // own H2 tile, for dW3
    pair_sync(cnt, expect, lane);

    // ---- loss and d(loss)/d(out) (both waves compute it; wave 0 keeps the statistics) ----
    float dout[O];
    {
      float out[O];
#pragma unroll
      for (int o = 0; o < O; ++o) out[o] = headp[o * 32 + j] + headp[(8 + o) * 32 + j] + sp[L::B3 + o];
      if constexpr (IS_PF) {
        const float lp_old = in_b;
        const float advn = valid ? (in_a - adv_mu) * adv_rstd : 0.0f;
        float zc[O], inv_var[O];
        float lp = 0.0f;
#pragma unroll
        for (int o = 0; o < O; ++o) {
          const float ls = fminf(fmaxf(sp[L::LS + o], -20.0f), 2.0f);   // continuous_policy.py:8-9,185
          inv_var[o] = __expf(-2.0f * ls);
          const float act = valid ? in_act[o] : 0.0f;
          lp += gauss_logp_term(act, out[o], inv_var[o], ls, a.tanh_action, zc[o]);
        }
        const float ratio = __expf(lp - lp_old);
        const float s1 = ratio * advn;
        const float s2 = fminf(fmaxf(ratio, 1.0f - a.clip_para), 1.0f + a.clip_para) * advn;
        const float g_lp = (valid && s1 <= s2) ? -advn * ratio * inv_b : 0.0f;
#pragma unroll
        for (int o = 0; o < O; ++o) {
          dout[o] = g_lp * zc[o] * inv_var[o];
          if (stat_lane && valid) {
            const float raw = sp[L::LS + o];                    // clamp passes gradient inside [-20, 2] only
            const float pass = (raw >= -20.0f && raw <= 2.0f) ? 1.0f : 0.0f;
            st[(7 + A + o) * 32 + j] += pass * (g_lp * (zc[o] * zc[o] * inv_var[o] - 1.0f) - a.entropy_coeff * inv_b);
          }
        }
#ifndef TRL_EXP_NOSTATS
        if (stat_lane && valid) {
          st[0 * 32 + j] += lp; st[1 * 32 + j] = fmaf(lp, lp, st[1 * 32 + j]); st[6 * 32 + j] -= fminf(s1, s2);
          st[2 * 32 + j] = fmaxf(st[2 * 32 + j], lp); st[3 * 32 + j] = fmaxf(st[3 * 32 + j], -lp);
          st[4 * 32 + j] = fmaxf(st[4 * 32 + j], ratio); st[5 * 32 + j] = fmaxf(st[5 * 32 + j], -ratio);
        }
#endif
      } else {
        const float v = out[0];
        const float R = in_a;
        float dv, l;
        if (a.clipped_value_loss) {                            // ppo.py:104-111
          const float vo = in_b;
          const float dc = v - vo;
          const float vc = vo + fminf(fmaxf(dc, -a.clip_para), a.clip_para);
          const float l1 = (v - R) * (v - R), l2 = (vc - R) * (vc - R);
          const float w1 = l1 > l2 ? 1.0f : (l1 == l2 ? 0.5f : 0.0f), w2 = 1.0f - w1;
          const float pass = (dc >= -a.clip_para && dc <= a.clip_para) ? 1.0f : 0.0f;
          l = 0.5f * fmaxf(l1, l2);
          dv = inv_b * (w1 * (v - R) + w2 * pass * (vc - R));
        } else {                                               // nn.MSELoss, a2c.py:43
          l = (v - R) * (v - R);
          dv = 2.0f * (v - R) * inv_b;
        }
        dout[0] = valid ? dv : 0.0f;
        if (stat_lane && valid) st[6 * 32 + j] += l;
      }
    }
    if (stat_lane) {
#pragma unroll
      for (int o = 0; o < O; ++o) st[(7 + o) * 32 + j] += dout[o];
#pragma unroll
      for (int o = 0; o < 4 * NQ; ++o) douts[j * 8 + o] = (o < O) ? dout[o] : 0.0f;
    }

    // ---- backward through the head, own features: dZ2 = (W3^T dout) * act'(H2) ----
    f32x16 dz2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
      for (int o = 0; o < O; ++o) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(sp + L::W3 + o * H + 32 * mo + 8 * q + 4 * hi);
        d0 = fmaf(w[0], dout[o], d0); d1 = fmaf(w[1], dout[o], d1);
        d2 = fmaf(w[2], dout[o], d2); d3 = fmaf(w[3], dout[o], d3);
      }
      dz2[4 * q + 0] = d0 * act_grad<ACT>(h2[4 * q + 0]); dz2[4 * q + 1] = d1 * act_grad<ACT>(h2[4 * q + 1]);
      dz2[4 * q + 2] = d2 * act_grad<ACT>(h2[4 * q + 2]); dz2[4 * q + 3] = d3 * act_grad<ACT>(h2[4 * q + 3]);
    }
    tile_store_T1(DZ2s, mo, dz2, j, hi);
    pair_sync(cnt, expect, lane);

    // X with lane = input feature, reg r = sample rowmap(r, hi) (68-byte coalesced row segments): issued
    // here, consumed by the dW1 MFMAs at the end of the tile
    float xn[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int sj = rowmap(r, hi);
      int64_t pr;
      if (contig) pr = pos - j + sj;                          // N % 32 == 0: the tile is one contiguous run of cells
      else        pr = __shfl(pos, sj, 64);
      const float v = a.obs[((s0 + sj < B) ? pr : 0) * D + (i < D ? i : 0)];
      xn[r] = (i < D && s0 + sj < B) ? v : 0.0f;
    }
    // ---- dW3[o][own f] += sum_s dout[o][s] H2[s][f]  (lane = feature) ----
    {
      const f32x16 h2n = tile_load_N(P, 0, i, hi);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const f32x4 d = *reinterpret_cast<const f32x4*>(douts + rowmap(r, hi) * 8 + 4 * q);
#pragma unroll
          for (int c = 0; c < 4; ++c) if (4 * q + c < O) gW3[4 * q + c] = fmaf(h2n[r], d[c], gW3[4 * q + c]);
        }
      }
    }
    // ---- dH1^T (own rows) = W2^T dZ2^T ; dZ1 = dH1 * act'(H1) ----
    f32x16 dz1;
    {
      const f32x16 dz2x = tile_load_T1(DZ2s, mx, j, hi);
      f32x16 acc = zero_tile();
      acc = layer_tile_wT_1src<L::LD2>(acc, sp + L::W2, mo, mo, dz2, i, hi);
      acc = layer_tile_wT_1src<L::LD2>(acc, sp + L::W2, mo, mx, dz2x, i, hi);
#pragma unroll
      for (int r = 0; r < 16; ++r) dz1[r] = acc[r] * act_grad<ACT>(h1[r]);
    }
    wave_lds_sync();                                          // P: H2 reads above precede the dZ1 writes below
    tile_store_T1(P, 0, dz1, j, hi);
    wave_lds_sync();

    // ---- dW2^T[own j_out][k_in] += sum_s dZ2[s][j_out] H1[s][k_in] ----
    {
      const f32x16 dzn = tile_load_N(DZ2s, mo, i, hi);
      float bs = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) bs += dzn[r];
      gb2 += bs;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          gW2[mb] = mfma32(dzn[r], H1s[(32 * mb + i) * TRL_TLD + rowmap(r, hi)], gW2[mb]);
    }
    // ---- dW1^T[own j_out][k_in] += sum_s dZ1[s][j_out] X[s][k_in] ----
    {
      const f32x16 dzn = tile_load_N(P, 0, i, hi);
      float bs = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) bs += dzn[r];
      gb1 += bs;
#pragma unroll
      for (int r = 0; r < 16; ++r) gW1 = mfma32(dzn[r], xn[r], gW1);
    }
    wave_lds_sync();
  }

  // ---- fold the 8 waves in fixed order into one partial gradient (flat layout) ----
  const int i = i0, hi = hi0, mo = mo0;
  // pull this pair's statistic slots into registers before the scratch area is recycled
  float stv[7], db3[O], dls[O];
  {
    const float* st = lds + PARF + pair * S::PAIR_SCR + S::O_ST;
#pragma unroll
    for (int k = 0; k < 7; ++k) stv[k] = st[k * 32 + i];
#pragma unroll
    for (int o = 0; o < O; ++o) { db3[o] = st[(7 + o) * 32 + i]; dls[o] = st[(7 + A + o) * 32 + i]; }
  }
  __syncthreads();
  float* gacc = lds + PARF;                                   // reuse scratch: S::P_STRIDE floats
  for (int e = tid; e < S::P_STRIDE; e += PPO_THREADS) gacc[e] = 0.0f;
  __syncthreads();
  for (int w = 0; w < PPO_WAVES; ++w) {
    if (wave == w) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int jo = 32 * mo + rowmap(r, hi);
        if (i < D) gacc[F::W1 + jo * D + i] += gW1[r];
        gacc[F::W2 + jo * H + i] += gW2[0][r];
        gacc[F::W2 + jo * H + 32 + i] += gW2[1][r];
      }
      // per-feature partials live in both lane halves: fold with one shuffle
      const float b1v = gb1 + __shfl_xor(gb1, 32, 64);
      const float b2v = gb2 + __shfl_xor(gb2, 32, 64);
      if (hi == 0) { gacc[F::B1 + 32 * mo + i] += b1v; gacc[F::B2 + 32 * mo + i] += b2v; }
#pragma unroll
      for (int o = 0; o < O; ++o) {
        const float w3v = gW3[o] + __shfl_xor(gW3[o], 32, 64);
        if (hi == 0) gacc[F::W3 + o * H + 32 * mo + i] += w3v;
      }
      if (mo == 0) {
#pragma unroll
        for (int o = 0; o < O; ++o) {
          const float b3v = wave_sum(hi == 0 ? db3[o] : 0.0f);
          if (lane == 0) gacc[F::B3 + o] += b3v;
          if (IS_PF) { const float lv = wave_sum(hi == 0 ? dls[o] : 0.0f); if (lane == 0) gacc[F::LS + o] += lv; }
        }
      }
    }
    __syncthreads();
  }
  const int wg = blockIdx.x;
  for (int e = tid; e < S::P_STRIDE; e += PPO_THREADS) a.partial[(size_t)wg * a.p_stride + e] = gacc[e];

  // ---- scalar statistics: wave shuffle reduce, then across the pairs through LDS ----
  __syncthreads();
  double* sred = reinterpret_cast<double*>(lds + PARF);
  if (mo == 0) {
    const bool own = hi == 0;
    const double v0 = wave_sum(own ? (double)stv[0] : 0.0), v1 = wave_sum(own ? (double)stv[1] : 0.0),
                 v6 = wave_sum(own ? (double)stv[6] : 0.0);
    const float v2 = wave_max(own ? stv[2] : -INFINITY), v3 = wave_max(own ? stv[3] : -INFINITY),
                v4 = wave_max(own ? stv[4] : -INFINITY), v5 = wave_max(own ? stv[5] : -INFINITY);
    if (lane == 0) {
      double* p = sred + pair * 8;
      p[0] = v0; p[1] = v1; p[2] = v2; p[3] = v3; p[4] = v4; p[5] = v5; p[6] = v6; p[7] = 0.0;
    }
  }
  __syncthreads();
  if (tid < 8) {
    double r = sred[tid];
    for (int w = 1; w < PPO_PAIRS; ++w) {
      const double o = sred[w * 8 + tid];
      r = (tid >= 2 && tid <= 5) ? fmax(r, o) : r + o;
    }
    a.scal_partial[(size_t)wg * 8 + tid] = r;
  }
}

template <int D, int H, int A, int ACT>
__global__ __launch_bounds__(PPO_THREADS, 2) void ppo_grad_kernel(PpoDev a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int half = a.n_wg >> 1;
  if ((int)blockIdx.x < half) ppo_net_pass<D, H, A, ACT, true>(a, lds, blockIdx.x, half);
  else                        ppo_net_pass<D, H, A, ACT, false>(a, lds, blockIdx.x - half, half);
}

// ---------------------------------------------------------------- partial reduce
// grads[p] = sum_w partial[w][p] in fixed order; info[] from the scalar partials:
//  0 policy surrogate sum (-min(s1,s2))   1 sum logp   2 sum logp^2   3 max logp   4 -min logp
//  5 max ratio   6 -min ratio   7 value-loss sum
#define RED_CHUNK 64
__global__ __launch_bounds__(256) void ppo_reduce_kernel(const float* __restrict__ partial,
                                                         const double* __restrict__ scal, int n_wg,
                                                         int p_stride, int p_pf, int p_vf,
                                                         const float* __restrict__ logstd, int n_act,
                                                         float* __restrict__ grads, double* __restrict__ info) {
  // block = 64 consecutive parameters x 4 waves; wave w folds partials w, w+4, ... with 4
  // independent accumulators (fixed order => deterministic), then the 4 waves fold through LDS.
  __shared__ float s_acc[4][RED_CHUNK];
  const int half = n_wg >> 1;
  const int net = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = blockIdx.x * RED_CHUNK + lane;
  const int pn = net == 0 ? p_pf : p_vf;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (p < pn) {
    const float* src = partial + (size_t)(net * half) * p_stride + p;
    int w = wave;
    for (; w + 12 < half; w += 16) {
      a0 += src[(size_t)w * p_stride];        a1 += src[(size_t)(w + 4) * p_stride];
      a2 += src[(size_t)(w + 8) * p_stride];  a3 += src[(size_t)(w + 12) * p_stride];
    }
    for (; w < half; w += 4) a0 += src[(size_t)w * p_stride];
  }
  s_acc[wave][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (wave == 0 && p < pn)
    grads[(net == 0 ? 0 : p_pf) + p] = (s_acc[0][lane] + s_acc[1][lane]) + (s_acc[2][lane] + s_acc[3][lane]);
  // scalar statistics: one wave per network, lanes stride over the workgroup partials (independent
  // loads, shuffle reduction) -- a serial 128-deep dependent-load chain here cost 50 us
  if (blockIdx.x == 0 && blockIdx.y == 0 && wave < 2) {
    const double* base = scal + (size_t)(wave * half) * 8;
    double v[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) v[k] = (k >= 2 && k <= 5) ? -INFINITY : 0.0;
    for (int w = lane; w < half; w += 64) {
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const double o = base[(size_t)w * 8 + k];
        v[k] = (k >= 2 && k <= 5) ? fmax(v[k], o) : v[k] + o;
      }
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) v[k] = (k >= 2 && k <= 5) ? wave_max(v[k]) : wave_sum(v[k]);
    if (lane == 0) {
      if (wave == 0) {
        info[0] = v[6]; info[1] = v[0]; info[2] = v[1]; info[3] = v[2]; info[4] = v[3]; info[5] = v[4]; info[6] = v[5];
        if (logstd) {                                        // log_std/{mean,std,max,min} (ppo.py:82-85)
          double sm = 0, sq = 0, mx = -INFINITY, mn = INFINITY;
          for (int o = 0; o < n_act; ++o) {
            const double x = fmin(fmax((double)logstd[o], -20.0), 2.0);
            sm += x; sq += x * x; mx = fmax(mx, x); mn = fmin(mn, x);
          }
          const double mean = sm / n_act;
          info[8] = mean;
          info[9] = n_act > 1 ? sqrt(fmax((sq - sm * mean) / (n_act - 1), 0.0)) : NAN;
          info[10] = mx; info[11] = mn;
        }
      } else {
        info[7] = v[6];
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
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int e = a.off[g] + tid;
    for (; e + 768 < a.off[g + 1]; e += 1024) {
      const float x0 = a.grads[e], x1 = a.grads[e + 256], x2 = a.grads[e + 512], x3 = a.grads[e + 768];
      s0 = fmaf(x0, x0, s0); s1 = fmaf(x1, x1, s1); s2 = fmaf(x2, x2, s2); s3 = fmaf(x3, x3, s3);
    }
    for (; e < a.off[g + 1]; e += 256) { const float x = a.grads[e]; s0 = fmaf(x, x, s0); }
    float ss = wave_sum((s0 + s1) + (s2 + s3)) * a.grad_scale * a.grad_scale;
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
  hipLaunchKernelGGL(ppo_reduce_kernel, dim3(trl_ceil_div(ps, RED_CHUNK), 2), dim3(256), 0, (hipStream_t)stream,
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
