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
#include <cstdlib>
#include "trl_common.h"
#include "trl_mlp.h"
#include "trl_comm.h"

struct PpoDev {
  const float *obs, *acts, *advs, *rets, *old_values, *old_logp;
  const int64_t* row_idx;
  int rows_mb, N;
  const double* adv_raw;
  double n_global;
  const float *pf_params, *vf_params;
  float clip_para, entropy_coeff;
  int clipped_value_loss, tanh_action;
  int loss_mode;                              // TRL_LOSS_PPO_CLIP (ppo.py:41-91) or TRL_LOSS_A2C (a2c.py:45-75)
  float* partial;
  double* scal_partial;
  int n_wg, n_pf, p_stride;                  // workgroups [0, n_pf) run the policy, [n_pf, n_wg) the value net
  int D, A;                                  // actual input / action dims (runtime-dims instantiation of the kernel)
};

template <int D, int H, int A> struct PpoShape {
  static_assert(H == 64 && D > 16 && D <= 32 && A <= 8, "instantiated for 16 < D <= 32, H == 64, A <= 8");
  static constexpr int P_PF = MlpFlat<D, H, A>::P_PF, P_VF = MlpFlat<D, H, 1>::P_VF;
  static constexpr int P_STRIDE = ((P_PF > P_VF ? P_PF : P_VF) + 63) & ~63;   // floats per workgroup partial
};

// ================================================================ one wave = one tile
// Every wave walks whole 16-sample tiles on its own: all four 16-feature slices of a layer are
// independent accumulator chains in one instruction stream (the matrix pipe always has an MFMA that
// does not wait on the previous one), layer outputs chain as the next layer's B operand in
// registers, and the only LDS traffic is (a) the shared read-only W2 / W2^T operand copies and
// (b) the wave-private transposes that put the sample index on the K axis for the weight-gradient
// GEMMs.  No inter-wave synchronisation inside the tile loop.
//   per tile (policy):  L1 20 + L2 64 + head 16 + dH2 16 + dH1 64 + dW3 16 + dW2 64 + dW1 16 = 276 MFMA
//   per tile (value):   L1 20 + L2 64 +                      dH1 64 +          dW2 64 + dW1 16 = 228 MFMA
// (bias gradients, the 17th input column of dW1 and the 1-wide value head are per-lane FMAs).
#define WV_WAVES 4
// development aid (tools/time_grad.py): per-phase cycle totals of every wave 0 into the spare tail of `partial`
#ifdef TRL_EXP_CLK
#define WCLK_DECL long long clk_prev = clock64(); float clk_acc[24]; for (int q_ = 0; q_ < 24; ++q_) clk_acc[q_] = 0.f;
#define WCLK(ph) { __builtin_amdgcn_sched_barrier(0); const long long c_ = clock64(); clk_acc[ph] += (float)(c_ - clk_prev); clk_prev = c_; __builtin_amdgcn_sched_barrier(0); }
#else
#define WCLK_DECL
#define WCLK(ph)
#endif
#define WV_THREADS (64 * WV_WAVES)
#define LDT 20                                   // staging row stride: 16 samples + pad, rows 16-B aligned
#define LDW 68                                   // W2 operand copies: 64 + 4, ds_read_b128 rows

template <int D, int H, int A> struct WvShape {
  static_assert(H == 64 && (D == 17 || D == 32) && A <= 8, "instantiated for D == 17 / 32, H == 64, A <= 8");
  // shared: W2 (row-major) | W2^T | b1 | b2 ; per wave: H1^T | H2^T (later dZ1^T) | dZ2^T | dout^T[16]
  static constexpr int O_W2F = 0, O_W2B = H * LDW, O_B1 = 2 * H * LDW, O_B2 = O_B1 + H, O_SCR = O_B2 + H;
  static constexpr int O_H1 = 0, O_H2 = H * LDT, O_DZ2 = 2 * H * LDT, O_DO = 3 * H * LDT, WSCR = O_DO + 16 * LDT;
  static constexpr int P_STRIDE = PpoShape<D, H, A>::P_STRIDE;
  static constexpr int MAIN = O_SCR + WV_WAVES * WSCR, FOLD = WV_WAVES * P_STRIDE;
  static constexpr int LDS_FLOATS = MAIN > FOLD ? MAIN : FOLD;
};

__device__ __forceinline__ f32x4 lds4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
// sum over the 16 lanes of a DPP row (same lane group g, all sample lanes j) with row rotations on the
// VALU (no LDS round trips); every lane ends with the total, callers read lane j == 0
__device__ __forceinline__ float row_sum16(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));  // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false));  // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xF, 0xF, false));  // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xF, 0xF, false));  // row_ror:1
  return v;
}

// RT = false: the network dims are the template's (the benchmark shape, everything folds to constants).
// RT = true: D and A are UPPER BOUNDS (17, 8) and the actual dims come from the launch (a.D in [2, 17], a.A in [1, 8]):
// the same 16 + 1-feature tile with the missing input features and outputs masked to zero where they are loaded, the flat
// parameter offsets and row strides computed from the actual dims -- Hopper / Swimmer / Walker / Reacher-shaped tasks
// (torchrl/networks/base.py:8-44 is shape-generic) stay on the two-launch path.
// D = 32 (WIDE, RT only; a.D in [18, 32] -- Ant's 27 observations): input features 16..31 are a SECOND 16-wide k group of
// the first layer (four more MFMA steps forward, a second accumulator tile per slice for dW1) instead of the single
// 17th column that rides on the VALU.
// WT: the workgroup's partial row is folded by other workgroups of THIS launch (the one-launch step): it leaves as
// device-scope write-through stores instead of plain ones
template <int D, int H, int A, int ACT, bool IS_PF, bool CONTIG, bool RT, bool WT = false>
__device__ void ppo_wave_pass(const PpoDev& a, float* lds, int wg_in_net, int n_wg_net) {
  using S = WvShape<D, H, A>;
  constexpr bool WIDE = D > 17;                      // a second 16-feature group (features 16 .. 31) on the matrix pipe
  static_assert(!WIDE || RT, "the wide tile exists as a runtime-dims instantiation only");
  const int Dr = RT ? a.D : D;                       // input features = row stride of obs and of W1
  const int O = IS_PF ? (RT ? a.A : A) : 1;          // outputs
  // offsets inside the flat parameter block for the actual dims (MlpFlat, trl_mlp.h)
  const int F_W1 = 0, F_B1 = H * Dr, F_W2 = F_B1 + H, F_B2 = F_W2 + H * H, F_W3 = F_B2 + H, F_B3 = F_W3 + O * H,
            F_LS = F_B3 + O, F_END = F_LS + O;
  const int PS = RT ? a.p_stride : S::P_STRIDE;      // floats of a partial row (the LDS images keep the template's stride)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4, i = j;
  const float* gp = IS_PF ? a.pf_params : a.vf_params;
  WCLK_DECL
  float* W2F = lds + S::O_W2F;
  float* W2B = lds + S::O_W2B;
  float* scr = lds + S::O_SCR + wave * S::WSCR;
  float* H1S = scr + S::O_H1;
  float* H2S = scr + S::O_H2;                       // H2^T until dW3 is done, then dZ1^T
  float* DZ2S = scr + S::O_DZ2;
  float* DOS = scr + S::O_DO;

  // The first tile's inputs sit behind a dependent pair of loads (row index -> cells): they are requested before
  // anything else, so that they travel while the weights are staged.
  // (requested before anything else: behind the set-up's barrier these two loads were a memory round trip of their own --
  // a load cannot be hoisted across a barrier by the compiler)
  const double adv_s1 = IS_PF ? a.adv_raw[0] : 0.0, adv_s2 = IS_PF ? a.adv_raw[1] : 0.0;
  const int B = a.rows_mb * a.N;
  const int n_tiles = (B + 15) / 16;
  const int tile_stride = n_wg_net * WV_WAVES;

  // Inputs are fetched ONE TILE AHEAD as raw values (masks are applied when they are consumed, so no wait
  // sits next to the loads) and the minibatch row index TWO tiles ahead (the dependent load behind it is
  // then off the critical path): x operand (5), X^T for dW1 (4), loss inputs (6 / 2).
  float xq[5], xtq[4], lq[6];
  float xq2[4], xtq2[4];                             // WIDE: features 16 + 4g + q of the x operand / feature 16 + j of X^T
  int tile = wg_in_net * WV_WAVES + wave;
  // ---- CONTIG (N % 16 == 0, hence B % 16 == 0: every tile is 16 consecutive envs of ONE time row, all samples valid) ----
  // The tile position (row, column tile) and the row's base addresses are wave-uniform: they live in scalar registers
  // and are advanced incrementally (no per-lane division, no 64-bit per-lane address arithmetic); each lane adds a
  // constant 32-bit byte offset, so a load is `global_load_dword v, v_off, s[base]` with nothing to compute in front.
  const int tpr = a.N >> 4;                                           // tiles per time row
  int p1r = 0, p1c = 0, p2r = 0, p2c = 0, st_r = 0, st_c = 0;         // positions of tile t+1 / t+2 (in strides), stride split
  int64_t ridx1 = 0, ridx2 = 0;
  // which of the tile's input features exist (RT): features 4g + q of the lane's x operand, feature 16, and feature j of
  // the lane's X^T operand; a missing feature is read from an in-range address and replaced by zero when it is consumed
  bool fv[4], fv2[4];
  const bool f16 = !WIDE && (!RT || Dr > 16), jv = !RT || i < Dr, jv2 = WIDE && 16 + i < Dr;
  const unsigned ox16 = (unsigned)(j * Dr + (f16 ? 16 : 0)) * 4u, os = (unsigned)j * 4u;
  unsigned oxq[4], oxt[4], oa_[4], oxq2[4], oxt2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    fv[q] = !RT || 4 * g + q < Dr;
    fv2[q] = WIDE && 16 + 4 * g + q < Dr;
    oxq[q] = (unsigned)(j * Dr + (fv[q] ? 4 * g + q : 0)) * 4u;
    oxt[q] = (unsigned)((4 * g + q) * Dr + (jv ? i : 0)) * 4u;
    oxq2[q] = (unsigned)(j * Dr + (fv2[q] ? 16 + 4 * g + q : 0)) * 4u;
    oxt2[q] = (unsigned)((4 * g + q) * Dr + (jv2 ? 16 + i : 0)) * 4u;
    oa_[q] = (unsigned)(j * O + (4 * g + q < O ? 4 * g + q : 0)) * 4u;
  }
  auto ldb = [](const float* base, unsigned byte_off) -> float {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
  };
  auto pos_advance = [&](int& r, int& c) { c += st_c; r += st_r; if (c >= tpr) { c -= tpr; ++r; } };
  auto load_ridx = [&](int r) -> int64_t {
    const int rr = r < a.rows_mb ? r : 0;                             // tiles past the end: any in-range row
    return a.row_idx ? a.row_idx[rr] : (int64_t)rr;
  };
  auto fetch_contig = [&](int64_t ridx, int r, int c) {
    const int64_t cell0 = (r < a.rows_mb) ? ridx * a.N + 16 * c : 0;  // wave-uniform
    const float* ob = a.obs + cell0 * Dr;
#pragma unroll
    for (int q = 0; q < 4; ++q) xq[q] = ldb(ob, oxq[q]);
    xq[4] = ldb(ob, ox16);
#pragma unroll
    for (int q = 0; q < 4; ++q) xtq[q] = ldb(ob, oxt[q]);           // X[sample 4g + q][feature j]: B operand of the dW1 GEMM
    if constexpr (WIDE) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { xq2[q] = ldb(ob, oxq2[q]); xtq2[q] = ldb(ob, oxt2[q]); }
    }
    if constexpr (IS_PF) {
      const float* ab = a.acts + cell0 * O;
#pragma unroll
      for (int q = 0; q < 4; ++q) lq[q] = ldb(ab, oa_[q]);
      lq[4] = ldb(a.advs + cell0, os);
      lq[5] = a.old_logp ? ldb(a.old_logp + cell0, os) : 0.0f;
    } else {
      lq[0] = lq[1] = lq[2] = lq[3] = 0.0f;
      lq[4] = ldb(a.rets + cell0, os);
      lq[5] = a.clipped_value_loss ? ldb(a.old_values + cell0, os) : 0.0f;
    }
  };
  // ---- generic path: any N, ragged last tile; per-lane cell indices ----
  auto row_of = [&](int smp, int& e) -> int { const int sm = smp < B ? smp : 0; const int r = sm / a.N; e = sm - r * a.N; return r; };
  auto fetch_inputs = [&](int64_t p, int s0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) xq[r] = a.obs[p * Dr + (fv[r] ? 4 * g + r : 0)];
    xq[4] = a.obs[p * Dr + (f16 ? 16 : 0)];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int sj = 4 * g + q;
      const int64_t pr = __shfl(p, sj, 64);
      xtq[q] = a.obs[(s0 + sj < B ? pr : 0) * Dr + (jv ? i : 0)];
      if constexpr (WIDE) {
        xq2[q] = a.obs[p * Dr + (fv2[q] ? 16 + 4 * g + q : 0)];
        xtq2[q] = a.obs[(s0 + sj < B ? pr : 0) * Dr + (jv2 ? 16 + i : 0)];
      }
    }
    if constexpr (IS_PF) {
#pragma unroll
      for (int r = 0; r < 4; ++r) lq[r] = a.acts[p * O + (4 * g + r < O ? 4 * g + r : 0)];
      lq[4] = a.advs[p]; lq[5] = a.old_logp ? a.old_logp[p] : 0.0f;
    } else {
      lq[0] = lq[1] = lq[2] = lq[3] = 0.0f;
      lq[4] = a.rets[p]; lq[5] = a.clipped_value_loss ? a.old_values[p] : 0.0f;
    }
  };
  int e_nn = 0;
  int64_t ridx_nn = 0;
  auto fetch_row = [&](int t) {
    const int r = row_of(t * 16 + j, e_nn);
    ridx_nn = a.row_idx ? a.row_idx[r] : (int64_t)r;
  };
  // loads of tile `nt` (= current + stride) and the row index of the tile after it
  auto prefetch_next = [&](int nt) {
    if constexpr (CONTIG) {
      fetch_contig(ridx1, p1r, p1c);
      p1r = p2r; p1c = p2c; ridx1 = ridx2;
      pos_advance(p2r, p2c);
      ridx2 = load_ridx(p2r);
    } else {
      const int64_t pn = (nt * 16 + j < B) ? ridx_nn * a.N + e_nn : 0;
      fetch_inputs(pn, nt * 16);
      fetch_row(nt + tile_stride);
    }
  };
  if constexpr (CONTIG) {
    const int t0 = __builtin_amdgcn_readfirstlane(tile);
    const int ts = __builtin_amdgcn_readfirstlane(tile_stride);
    st_r = ts / tpr; st_c = ts - st_r * tpr;
    int r0 = t0 / tpr, c0 = t0 - (t0 / tpr) * tpr;
    fetch_contig(load_ridx(r0), r0, c0);                              // tile t (dependent pair of loads, once per kernel)
    p1r = r0; p1c = c0;
    pos_advance(p1r, p1c);
    ridx1 = load_ridx(p1r);
    p2r = p1r; p2c = p1c;
    pos_advance(p2r, p2c);
    ridx2 = load_ridx(p2r);
  } else {
    fetch_row(tile);
    const int64_t p0 = (tile * 16 + j < B) ? ridx_nn * a.N + e_nn : 0;
    fetch_inputs(p0, tile * 16);
    fetch_row(tile + tile_stride);
  }
  // ---- one-time setup ----
  constexpr int NV = H * H / (4 * WV_THREADS);    // all loads in flight before the first LDS store
  f32x4 wv[NV], wt[NV];
  const int tc = tid & 63, tq = tid >> 6;
  const float bias1 = tid < H ? gp[F_B1 + tid] : 0.0f, bias2 = tid < H ? gp[F_B2 + tid] : 0.0f;
  {
    // W2 is staged twice, row-major (forward A operand: W2[own row][k]) and transposed (backward A operand: W2[k][own
    // column]).  Both copies are written with 16-byte LDS stores: the transposed one from a second, column-wise read of
    // the 16 KB block (coalesced along the column index, served by L1 / L2) -- scattering the row-wise registers instead
    // took 64 four-byte stores per thread whose 16 lanes of a row hit two banks.
    // Every global load of the set-up (these, and the register-resident weight slices below) is requested before the
    // first LDS store waits for one of them.
    static_assert(WV_THREADS == 256 && H == 64, "the transposed staging maps (column, 4 rows) onto 256 threads");
    // (runtime dims: the value block starts P_PF floats into the flat buffer, which is a multiple of 4 only for even A)
    const bool w2_al = !RT || ((reinterpret_cast<uintptr_t>(gp + F_W2) & 15) == 0);
#pragma unroll
    for (int t = 0; t < NV; ++t) {
      if (w2_al) wv[t] = *reinterpret_cast<const f32x4*>(gp + F_W2 + 4 * (tid + WV_THREADS * t));
      else {
#pragma unroll
        for (int x = 0; x < 4; ++x) wv[t][x] = gp[F_W2 + 4 * (tid + WV_THREADS * t) + x];
      }
    }
#pragma unroll
    for (int t = 0; t < NV; ++t)
#pragma unroll
      for (int x = 0; x < 4; ++x) wt[t][x] = gp[F_W2 + (4 * (tq + 4 * t) + x) * H + tc];       // W2[r4 + x][column tc]
  }
  for (int e = lane; e < 16 * LDT; e += 64) DOS[e] = 0.0f;       // dout rows >= O stay zero
  // register-resident A operands, lane (i, g): k index of MFMA step (slice sl, r) is feature 16 sl + 4 g + r
  float w1[4][5], w3h[4][4], w3t[4][4];
  float w1b[4][4];                                   // WIDE: W1[row][16 + 4g + r]
#pragma unroll
  for (int so = 0; so < 4; ++so) {
    const int row = 16 * so + i;
#pragma unroll
    for (int r = 0; r < 4; ++r) w1[so][r] = fv[r] ? gp[F_W1 + row * Dr + (fv[r] ? 4 * g + r : 0)] : 0.0f;
    w1[so][4] = (g == 0 && f16) ? gp[F_W1 + row * Dr + (f16 ? 16 : 0)] : 0.0f;
    if constexpr (WIDE) {
#pragma unroll
      for (int r = 0; r < 4; ++r) w1b[so][r] = fv2[r] ? gp[F_W1 + row * Dr + (fv2[r] ? 16 + 4 * g + r : 0)] : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if constexpr (IS_PF) {
        w3h[so][r] = (i < O) ? gp[F_W3 + (i < O ? i : 0) * H + 16 * so + 4 * g + r] : 0.0f;     // head: W3[o = i][k]
        const int o = 4 * g + r;
        w3t[so][r] = (o < O) ? gp[F_W3 + (o < O ? o : 0) * H + row] : 0.0f;                     // dH2: W3[k = o][own f]
      } else {
        w3h[so][r] = gp[F_W3 + 16 * so + 4 * g + r];                                            // per-lane head weights
        w3t[so][r] = 0.0f;
      }
    }
  }
  // per-lane constants of the lane's 4 outputs o = 4g + r (policy)
  float lsv[4], ivv[4], b3v[4], lspass[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = 4 * g + r;
    const float raw = (IS_PF && o < O) ? gp[F_LS + (o < O ? o : 0)] : 0.0f;
    lsv[r] = fminf(fmaxf(raw, -20.0f), 2.0f);                   // continuous_policy.py:8-9,185
    ivv[r] = __expf(-2.0f * lsv[r]);
    lspass[r] = (raw >= -20.0f && raw <= 2.0f) ? 1.0f : 0.0f;   // clamp passes gradient inside [-20, 2]
    b3v[r] = (o < O) ? gp[F_B3 + (o < O ? o : 0)] : 0.0f;
  }
  const float vb3 = IS_PF ? 0.0f : gp[F_B3];
#pragma unroll
  for (int t = 0; t < NV; ++t) {
    const int e = 4 * (tid + WV_THREADS * t), r = e / H, c = e - r * H;
    *reinterpret_cast<f32x4*>(W2F + r * LDW + c) = wv[t];
    *reinterpret_cast<f32x4*>(W2B + tc * LDW + 4 * (tq + 4 * t)) = wt[t];
  }
  if (tid < H) { lds[S::O_B1 + tid] = bias1; lds[S::O_B2 + tid] = bias2; }
  __syncthreads();

  // advantage normalisation constants (ppo.py:141-147): mean, unbiased std
  const double ng = a.n_global;
  const double adv_mean = adv_s1 / ng;
  const double adv_var = (adv_s2 - adv_s1 * adv_s1 / ng) / (ng - 1.0);
  const float adv_mu = (float)adv_mean;
  const float adv_rstd = 1.0f / ((float)sqrt(fmax(adv_var, 0.0)) + 1e-5f);
  const float inv_b = (float)(1.0 / ng);

  // ---- gradient accumulators ----
  f32x4 gW2[4][4], gW1[4], gW3[4];                  // MFMA tiles: dW2[f2 slice][f1 slice], dW1[f1 slice][k < 16], dW3
  f32x4 gW1b[4];                                     // WIDE: dW1[f1 slice][16 <= k < 32]
  float gb1[4][4], gb2[4][4], gW1c[4][4];           // per-lane (own sample) partials: db1, db2, dW1[:, 16]
  float db3[4], dls[4], stv[7];
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    gW1[x] = gW3[x] = gW1b[x] = f32x4{0.f, 0.f, 0.f, 0.f};
    db3[x] = dls[x] = 0.0f;
#pragma unroll
    for (int y = 0; y < 4; ++y) { gW2[x][y] = f32x4{0.f, 0.f, 0.f, 0.f}; gb1[x][y] = gb2[x][y] = gW1c[x][y] = 0.0f; }
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) stv[k] = (k >= 2 && k <= 5) ? -INFINITY : 0.0f;

  WCLK(0)
  for (; tile < n_tiles; tile += tile_stride) {
    const int s = tile * 16 + j;
    const bool valid = CONTIG ? true : (s < B);
    float xb[5], xt[4], lin[6], xb2[4], xt2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) xb[r] = (valid && fv[r]) ? xq[r] : 0.0f;
    xb[4] = (valid && f16) ? xq[4] : 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) xt[q] = (jv && (CONTIG || tile * 16 + 4 * g + q < B)) ? xtq[q] : 0.0f;
    if constexpr (WIDE) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        xb2[q] = (valid && fv2[q]) ? xq2[q] : 0.0f;
        xt2[q] = (jv2 && (CONTIG || tile * 16 + 4 * g + q < B)) ? xtq2[q] : 0.0f;
      }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) lin[q] = lq[q];
    const float x16 = xb[4];

    // LDS operand reads are issued one phase AHEAD of the MFMAs that consume them (PIN_DS keeps the compiler from
    // sinking them back next to their use): the ~130-cycle LDS round trip then runs under the previous phase's matrix
    // work instead of stalling the in-order wave in front of every 4-MFMA group.
#define PIN_DS() __builtin_amdgcn_sched_barrier(0x676)   /* VALU / SALU / VMEM / DS writes / transcendentals may cross; DS reads and MFMAs may not */
    // ---- forward layer 1: 4 independent slices ----
    f32x4 h1[4], h2[4];
    f32x4 wq[4], bq;                                   // W2 operand slices / b2 slice of the NEXT L2 output slice
#pragma unroll
    for (int so = 0; so < 4; ++so) h1[so] = lds4(lds + S::O_B1 + 16 * so + 4 * g);
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) wq[sl] = lds4(W2F + i * LDW + 16 * sl + 4 * g);
    bq = lds4(lds + S::O_B2 + 4 * g);
    PIN_DS();
    WCLK(9)
#pragma unroll
    for (int q = 0; q < (WIDE ? 4 : 5); ++q)
#pragma unroll
      for (int so = 0; so < 4; ++so) h1[so] = mfma16(w1[so][q], xb[q], h1[so]);
    if constexpr (WIDE) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int so = 0; so < 4; ++so) h1[so] = mfma16(w1b[so][q], xb2[q], h1[so]);
    }
    WCLK(10)
#pragma unroll
    for (int so = 0; so < 4; ++so)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        h1[so][r] = act_fn<ACT>(h1[so][r]);
        H1S[(16 * so + 4 * g + r) * LDT + j] = h1[so][r];        // H1^T[f][s] for dW2
      }
    WCLK(1)
    // ---- forward layer 2 ----
#pragma unroll
    for (int so = 0; so < 4; ++so) {
      f32x4 acc = bq;
      f32x4 wc[4];
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) wc[sl] = wq[sl];
      if (so < 3) {
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) wq[sl] = lds4(W2F + (16 * (so + 1) + i) * LDW + 16 * sl + 4 * g);
        bq = lds4(lds + S::O_B2 + 16 * (so + 1) + 4 * g);
        PIN_DS();
      }
#pragma unroll
      for (int sl = 0; sl < 4; ++sl)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = mfma16(wc[sl][r], h1[sl][r], acc);
      WCLK(11)
#pragma unroll
      for (int r = 0; r < 4; ++r) h2[so][r] = act_fn<ACT>(acc[r]);
      WCLK(12)
    }

    // Next tile's loads are issued HERE, mid-tile: at the top of the loop every outstanding load is then
    // half a tile old, and no wait next to the first MFMAs can stall on a load that was just issued.
    prefetch_next(tile + tile_stride);            // loads of tile t+1 (addresses were resolved a tile ago), row index of tile t+2
    WCLK(2)
    // ---- head, loss, d(loss)/d(out), dZ2 ----
    f32x4 dz2[4];
    f32x4 wb[4];                                       // W2^T operand slices of the next dH1 k-slice
    if constexpr (IS_PF) {
#pragma unroll
      for (int so = 0; so < 4; ++so)
#pragma unroll
        for (int r = 0; r < 4; ++r) H2S[(16 * so + 4 * g + r) * LDT + j] = h2[so][r];            // H2^T[f][s] for dW3
      // out^T[o][s]: two interleaved accumulation chains over the 64 features
      f32x4 oa = f32x4{b3v[0], b3v[1], b3v[2], b3v[3]}, ob = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sl = 0; sl < 4; sl += 2)
#pragma unroll
        for (int r = 0; r < 4; ++r) { oa = mfma16(w3h[sl][r], h2[sl][r], oa); ob = mfma16(w3h[sl + 1][r], h2[sl + 1][r], ob); }
      WCLK(14)
      // lane (j, g < 2) owns outputs o = 4g + r of sample j
      float zc[4], lp = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {                                  // branch-free: lanes without an output compute on zeros
        const bool has = 4 * g + r < O;
        const float t = gauss_logp_term((valid && has) ? lin[r] : 0.0f, has ? oa[r] + ob[r] : 0.0f, ivv[r], lsv[r],
                                        a.tanh_action, zc[r]);
        lp += has ? t : 0.0f;
        zc[r] = has ? zc[r] : 0.0f;
      }
      WCLK(18)
      lp += __shfl_xor(lp, 16, 64);                                // outputs 0..3 (g = 0) + 4..7 (g = 1)
      WCLK(19)
      const float advn = valid ? (lin[4] - adv_mu) * adv_rstd : 0.0f;
      float ratio, s1, s2, g_lp;
      if (a.loss_mode == TRL_LOSS_A2C) {                            // L = -mean(log pi * adv) (a2c.py:69-70)
        ratio = 1.0f;
        s1 = s2 = lp * advn;
        g_lp = (valid && g < 2) ? -advn * inv_b : 0.0f;
      } else {                                                     // clipped surrogate (ppo.py:58-66)
        ratio = __expf(lp - lin[5]);
        s1 = ratio * advn;
        s2 = fminf(fmaxf(ratio, 1.0f - a.clip_para), 1.0f + a.clip_para) * advn;
        g_lp = (valid && g < 2 && s1 <= s2) ? -advn * ratio * inv_b : 0.0f;
      }
      WCLK(20)
      float dout[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool own = valid && g < 2 && 4 * g + r < O;
        dout[r] = own ? g_lp * zc[r] * ivv[r] : 0.0f;
        db3[r] += dout[r];
        dls[r] += own ? lspass[r] * (g_lp * (zc[r] * zc[r] * ivv[r] - 1.0f) - a.entropy_coeff * inv_b) : 0.0f;
        if (g < 2) DOS[(4 * g + r) * LDT + j] = dout[r];           // dout^T[o][s] for dW3
      }
      WCLK(21)
      if (valid && g == 0) {
        stv[0] += lp; stv[1] = fmaf(lp, lp, stv[1]); stv[6] -= fminf(s1, s2);
        stv[2] = fmaxf(stv[2], lp); stv[3] = fmaxf(stv[3], -lp);
        stv[4] = fmaxf(stv[4], ratio); stv[5] = fmaxf(stv[5], -ratio);
      }
      WCLK(15)
      // operands of dW3 (and of the first dH1 k-slice) are requested now and arrive under the dH2 MFMAs
      const f32x4 da = lds4(DOS + i * LDT + 4 * g);
      f32x4 hb3[4];
#pragma unroll
      for (int so = 0; so < 4; ++so) hb3[so] = lds4(H2S + (16 * so + j) * LDT + 4 * g);
#pragma unroll
      for (int so = 0; so < 4; ++so) wb[so] = lds4(W2B + (16 * so + i) * LDW + 4 * g);
      PIN_DS();
      // dH2^T[f][s] = sum_o W3[o][f] dout[o][s];  dZ2 = dH2 * act'(H2)
#pragma unroll
      for (int so = 0; so < 4; ++so) dz2[so] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int so = 0; so < 4; ++so) dz2[so] = mfma16(w3t[so][r], dout[r], dz2[so]);
      // dW3[o][f] += sum_s dout[s][o] H2[s][f]   (K step q is sample 4g + q)
#pragma unroll
      for (int so = 0; so < 4; ++so)
#pragma unroll
        for (int q = 0; q < 4; ++q) gW3[so] = mfma16(da[q], hb3[so][q], gW3[so]);
      WCLK(16)
    } else {
#pragma unroll
      for (int so = 0; so < 4; ++so) wb[so] = lds4(W2B + (16 * so + i) * LDW + 4 * g);
      PIN_DS();
      float v = 0.0f;
#pragma unroll
      for (int so = 0; so < 4; ++so)
#pragma unroll
        for (int r = 0; r < 4; ++r) v = fmaf(w3h[so][r], h2[so][r], v);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      v += vb3;
      const float R = lin[4];
      float dv, l;
      if (a.clipped_value_loss) {                                  // ppo.py:104-111
        const float vo = lin[5];
        const float dc = v - vo;
        const float vc = vo + fminf(fmaxf(dc, -a.clip_para), a.clip_para);
        const float l1 = (v - R) * (v - R), l2 = (vc - R) * (vc - R);
        const float wa = l1 > l2 ? 1.0f : (l1 == l2 ? 0.5f : 0.0f), wb_ = 1.0f - wa;
        const float pass = (dc >= -a.clip_para && dc <= a.clip_para) ? 1.0f : 0.0f;
        l = 0.5f * fmaxf(l1, l2);
        dv = inv_b * (wa * (v - R) + wb_ * pass * (vc - R));
      } else {                                                     // nn.MSELoss, a2c.py:43
        l = (v - R) * (v - R);
        dv = 2.0f * (v - R) * inv_b;
      }
      dv = valid ? dv : 0.0f;
      if (valid && g == 0) {
        stv[6] += l; db3[0] += dv;
        stv[0] += v; stv[1] = fmaf(v, v, stv[1]); stv[2] = fmaxf(stv[2], v); stv[3] = fmaxf(stv[3], -v);   // v_pred/*
      }
#pragma unroll
      for (int so = 0; so < 4; ++so)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dz2[so][r] = dv * w3h[so][r];
          gW3[so][r] = fmaf(dv, h2[so][r], gW3[so][r]);             // per-lane dW3[f] partial (own sample)
        }
    }
#pragma unroll
    for (int so = 0; so < 4; ++so)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dz2[so][r] *= act_grad<ACT>(h2[so][r]);
        gb2[so][r] += dz2[so][r];
        DZ2S[(16 * so + 4 * g + r) * LDT + j] = dz2[so][r];        // dZ2^T[f][s] for dW2
      }

    WCLK(3)
    // ---- dH1^T (all slices) = W2^T dZ2^T ; dZ1 = dH1 * act'(H1) ----
    f32x4 dz1[4];
    f32x4 hb[4], za2;                                  // dW2 operands: H1 rows (all four slices), dZ2 rows of the next slice
#pragma unroll
    for (int so = 0; so < 4; ++so) dz1[so] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      f32x4 w[4];
#pragma unroll
      for (int so = 0; so < 4; ++so) w[so] = wb[so];
      if (sl < 3) {
#pragma unroll
        for (int so = 0; so < 4; ++so) wb[so] = lds4(W2B + (16 * so + i) * LDW + 16 * (sl + 1) + 4 * g);
      } else {                                         // last k-slice: the dW2 operands are requested under it
#pragma unroll
        for (int c = 0; c < 4; ++c) hb[c] = lds4(H1S + (16 * c + j) * LDT + 4 * g);
        za2 = lds4(DZ2S + i * LDT + 4 * g);
      }
      PIN_DS();
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int so = 0; so < 4; ++so) dz1[so] = mfma16(w[so][r], dz2[sl][r], dz1[so]);
    }
    WCLK(17)
#pragma unroll
    for (int so = 0; so < 4; ++so)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dz1[so][r] *= act_grad<ACT>(h1[so][r]);
        gb1[so][r] += dz1[so][r];
        if constexpr (!WIDE) gW1c[so][r] = fmaf(dz1[so][r], x16, gW1c[so][r]);
        H2S[(16 * so + 4 * g + r) * LDT + j] = dz1[so][r];         // dZ1^T[f][s] for dW1 (H2^T is consumed)
      }
    // the dW1 operands (dZ1 rows) are requested now and arrive under the dW2 MFMAs
    f32x4 za1[4];
#pragma unroll
    for (int so = 0; so < 4; ++so) za1[so] = lds4(H2S + (16 * so + i) * LDT + 4 * g);
    PIN_DS();

    WCLK(4)
    // ---- dW2[f2][f1] += sum_s dZ2[s][f2] H1[s][f1]  (K step q is sample 4g + q) ----
#pragma unroll
    for (int so = 0; so < 4; ++so) {
      const f32x4 za = za2;
      if (so < 3) { za2 = lds4(DZ2S + (16 * (so + 1) + i) * LDT + 4 * g); PIN_DS(); }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) gW2[so][c] = mfma16(za[q], hb[c][q], gW2[so][c]);
    }
    WCLK(5)
    // ---- dW1[f1][k < 16] += sum_s dZ1[s][f1] X[s][k] ----
#pragma unroll
    for (int so = 0; so < 4; ++so)
#pragma unroll
      for (int q = 0; q < 4; ++q) gW1[so] = mfma16(za1[so][q], xt[q], gW1[so]);
    if constexpr (WIDE) {
#pragma unroll
      for (int so = 0; so < 4; ++so)
#pragma unroll
        for (int q = 0; q < 4; ++q) gW1b[so] = mfma16(za1[so][q], xt2[q], gW1b[so]);
    }
#undef PIN_DS
    WCLK(6)
  }

  // ---- epilogue: every wave writes its gradient image, then all threads add the 4 images in fixed order ----
  __syncthreads();                                                 // staging / weight copies are dead from here
  float* gimg = lds + wave * S::P_STRIDE;            // every parameter slot below is written exactly once
  for (int e = F_END + lane; e < PS; e += 64) gimg[e] = 0.0f;              // padding (and logstd slots of the value net)
#pragma unroll
  for (int so = 0; so < 4; ++so)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = 16 * so + 4 * g + r;                           // output feature (row of W2 / W1)
#pragma unroll
      for (int c = 0; c < 4; ++c) gimg[F_W2 + f * H + 16 * c + j] = gW2[so][c][r];
      if (jv) gimg[F_W1 + f * Dr + j] = gW1[so][r];
      if (WIDE && jv2) gimg[F_W1 + f * Dr + 16 + j] = gW1b[so][r];
      const float c16 = WIDE ? 0.0f : row_sum16(gW1c[so][r]), s1 = row_sum16(gb1[so][r]), s2 = row_sum16(gb2[so][r]);
      if (j == 0) { if (f16) gimg[F_W1 + f * Dr + 16] = c16; gimg[F_B1 + f] = s1; gimg[F_B2 + f] = s2; }
      if constexpr (IS_PF) {
        const int o = 4 * g + r;                                   // gW3 rows are outputs
        if (o < O) gimg[F_W3 + o * H + 16 * so + j] = gW3[so][r];
      } else {
        const float w3s = row_sum16(gW3[so][r]);
        if (j == 0) gimg[F_W3 + f] = w3s;
      }
    }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = 4 * g + r;
    const float b3 = row_sum16(db3[r]), dl = row_sum16(dls[r]);
    if (j == 0 && o < O) {
      gimg[F_B3 + o] = b3;
      if (IS_PF) gimg[F_LS + o] = dl;
    }
  }
  WCLK(7)
  __syncthreads();
  const int wg = blockIdx.x;
  static_assert(S::P_STRIDE % 4 == 0, "partial rows are folded and stored 16 bytes at a time");
  for (int e4 = tid; e4 < PS / 4; e4 += WV_THREADS) {
    f32x4 acc = lds4(lds + 4 * e4);
#pragma unroll
    for (int w = 1; w < WV_WAVES; ++w) acc += lds4(lds + w * S::P_STRIDE + 4 * e4);          // same order per element
    if constexpr (WT) {      // read by other workgroups of THIS launch: written through to the device's coherence point
      unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.partial + (size_t)wg * a.p_stride + 4 * e4);
      __hip_atomic_store(dst, ((unsigned long long)__float_as_uint(acc[1]) << 32) | __float_as_uint(acc[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(dst + 1, ((unsigned long long)__float_as_uint(acc[3]) << 32) | __float_as_uint(acc[2]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      *reinterpret_cast<f32x4*>(a.partial + (size_t)wg * a.p_stride + 4 * e4) = acc;
    }
  }
  WCLK(8)
#ifdef TRL_EXP_CLK
  if (tid == 0) for (int ph = 0; ph < 24; ++ph) a.partial[(size_t)wg * a.p_stride + PS - 40 + ph] = clk_acc[ph];
#endif
  // ---- scalar statistics ----
  __syncthreads();
  double* sred = reinterpret_cast<double*>(lds);
  {
    const bool own = g == 0;
    const double v0 = wave_sum(own ? (double)stv[0] : 0.0), v1 = wave_sum(own ? (double)stv[1] : 0.0),
                 v6 = wave_sum(own ? (double)stv[6] : 0.0);
    const float v2 = wave_max(own ? stv[2] : -INFINITY), v3 = wave_max(own ? stv[3] : -INFINITY),
                v4 = wave_max(own ? stv[4] : -INFINITY), v5 = wave_max(own ? stv[5] : -INFINITY);
    if (lane == 0) {
      double* p = sred + wave * 8;
      p[0] = v0; p[1] = v1; p[2] = v2; p[3] = v3; p[4] = v4; p[5] = v5; p[6] = v6; p[7] = 0.0;
    }
  }
  __syncthreads();
  if (tid < 8) {
    double r = sred[tid];
    for (int w = 1; w < WV_WAVES; ++w) {
      const double o = sred[w * 8 + tid];
      r = (tid >= 2 && tid <= 5) ? fmax(r, o) : r + o;
    }
    if constexpr (WT)
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(a.scal_partial + (size_t)wg * 8 + tid),
                         (unsigned long long)__double_as_longlong(r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      a.scal_partial[(size_t)wg * 8 + tid] = r;
  }
}

// ---------------------------------------------------------------- partial reduce
// grads[p] = sum_w partial[w][p] in fixed order; info[] from the scalar partials:
//  0 policy surrogate sum (-min(s1,s2))   1 sum logp   2 sum logp^2   3 max logp   4 -min logp
//  5 max ratio   6 -min ratio   7 value-loss sum
#define RED_CHUNK 64
// waves per block of 64 parameters / partial rows a lane has in flight per round.  Measured on MI355X at 147 + 109 rows
// (rocprofv3 mean of the fused launch, round 3, after its argument struct stopped going through scratch memory):
// 4 x 16 -> 7.9 us, 8 x 16 -> 7.4 us, 8 x 32 -> 7.8 us, 16 x 16 -> 7.8 us (profiles/r03_grad_kernel_experiments.txt); the
// launch is bound by its rendezvous and launch latencies: without any partial-row load it still takes ~6 us.
#ifndef RED_WAVES
#define RED_WAVES 8
#endif
#ifndef RED_DEPTH
#define RED_DEPTH 16
#endif
// ---- pieces of the fold shared by the stand-alone launch (8 waves per 64 parameters) and by the tail of the fused
// minibatch step (4 waves, each playing two of the 8): the association order is part of the contract -- both give the
// same bits.  Rows of one network's partials are dealt to 8 "fold waves" (row r -> wave r % 8), each with RED_DEPTH
// independent chains (chain k takes rows w + 8 k + 128 round, in round order), chains folded pairwise (k with
// k + 8, 4, 2, 1), fold waves as ((0 + 1) + (2 + 3)) + ((4 + 5) + (6 + 7)).
__device__ __forceinline__ float fold_chain_tree(float (&acc)[RED_DEPTH]) {
#pragma unroll
  for (int st = RED_DEPTH / 2; st > 0; st >>= 1)
#pragma unroll
    for (int k = 0; k < st; ++k) acc[k] += acc[k + st];
  return acc[0];
}
// one fold wave: src = first row of the network + this lane's parameter
__device__ __forceinline__ float fold_rows_wave(const float* __restrict__ src, int nrow, int p_stride, int w, bool in_range) {
  // 16 independent loads in flight per lane and round: the fold is latency-, not bandwidth-bound
  float acc[RED_DEPTH];
#pragma unroll
  for (int k = 0; k < RED_DEPTH; ++k) acc[k] = 0.0f;
  if (in_range) {
    for (; w < nrow; w += RED_DEPTH * RED_WAVES) { // predicated: a ragged row count must not fall back to a serial tail
      float v[RED_DEPTH];
#pragma unroll
      for (int k = 0; k < RED_DEPTH; ++k) v[k] = (w + RED_WAVES * k < nrow) ? src[(size_t)(w + RED_WAVES * k) * p_stride] : 0.0f;
#pragma unroll
      for (int k = 0; k < RED_DEPTH; ++k) acc[k] += v[k];
    }
  }
  return fold_chain_tree(acc);
}
// two fold waves (w and w + 4) by one real wave, all their loads of a round in flight together
// COH: the rows were written by other workgroups of THIS launch -- device-coherent loads (not served from this XCD's L2)
template <bool COH> __device__ __forceinline__ float row_load(const float* p) {
  if constexpr (COH) return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  else return *p;
}
template <bool COH>
__device__ __forceinline__ void fold_rows_wave_pair(const float* __restrict__ src, int nrow, int p_stride, int w, bool in_range,
                                                    float& r0, float& r1) {
  float a0[RED_DEPTH], a1[RED_DEPTH];
#pragma unroll
  for (int k = 0; k < RED_DEPTH; ++k) a0[k] = a1[k] = 0.0f;
  if (in_range) {
    for (; w < nrow; w += RED_DEPTH * RED_WAVES) {
      float v0[RED_DEPTH], v1[RED_DEPTH];
#pragma unroll
      for (int k = 0; k < RED_DEPTH; ++k) {
        v0[k] = (w + RED_WAVES * k < nrow) ? row_load<COH>(src + (size_t)(w + RED_WAVES * k) * p_stride) : 0.0f;
        v1[k] = (w + 4 + RED_WAVES * k < nrow) ? row_load<COH>(src + (size_t)(w + 4 + RED_WAVES * k) * p_stride) : 0.0f;
      }
#pragma unroll
      for (int k = 0; k < RED_DEPTH; ++k) { a0[k] += v0[k]; a1[k] += v1[k]; }
    }
  }
  r0 = fold_chain_tree(a0);
  r1 = fold_chain_tree(a1);
}
__device__ __forceinline__ float fold_waves_tree(const float (*s_acc)[RED_CHUNK], int lane) {
  float t4[RED_WAVES / 4];
#pragma unroll
  for (int q = 0; q < RED_WAVES / 4; ++q)
    t4[q] = (s_acc[4 * q][lane] + s_acc[4 * q + 1][lane]) + (s_acc[4 * q + 2][lane] + s_acc[4 * q + 3][lane]);
  float gval = t4[0];
#pragma unroll
  for (int q = 1; q < RED_WAVES / 4; ++q) gval += t4[q];
  return gval;
}
// one wave: a network's loss / log-prob / value statistics from the workgroups' scalar rows, every row requested at once
// (3 x 7 loads in flight per lane; the per-lane accumulation order is the one of the strided loop)
template <bool COH = false>
__device__ __noinline__ void fold_scalar_stats(const double* __restrict__ base, int nrow, int net, int lane,
                                                  double* __restrict__ info) {
  double v[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) v[k] = (k >= 2 && k <= 5) ? -INFINITY : 0.0;
  // (three rows per lane and round: up to 192 workgroups per network in ONE round trip with 42 doubles in flight -- the
  // function is out of line and the launches that call it are bounded to 128 registers, see ppo_reduce_adam_kernel; a lane
  // still adds its rows lane, lane + 64, lane + 128, ... in that order)
  for (int w0 = 0; w0 < nrow; w0 += 192) {
    double o[3][7];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int w = w0 + lane + 64 * q;
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        double x = (k >= 2 && k <= 5) ? -INFINITY : 0.0;
        if (w < nrow) {
          if constexpr (COH) x = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(base + (size_t)w * 8 + k),
                                                                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          else x = base[(size_t)w * 8 + k];
        }
        o[q][k] = x;
      }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int k = 0; k < 7; ++k) v[k] = (k >= 2 && k <= 5) ? fmax(v[k], o[q][k]) : v[k] + o[q][k];
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) v[k] = (k >= 2 && k <= 5) ? wave_max(v[k]) : wave_sum(v[k]);
  if (lane == 0) {
    if (net == 0) {
      info[0] = v[6]; info[1] = v[0]; info[2] = v[1]; info[3] = v[2]; info[4] = v[3]; info[5] = v[4]; info[6] = v[5];
    } else {
      info[7] = v[6];
      info[12] = v[0]; info[13] = v[1]; info[14] = v[2]; info[15] = v[3];   // v_pred: sum, sum of squares, max, -min
    }
  }
}
// one wave: log_std/{mean,std,max,min} (ppo.py:82-85) and the same four for std = exp(clamped logstd) (a2c.py:95-100),
// one action dimension per lane (raw = the lane's log_std as stored, lanes >= n_act: anything)
// (not inlined, like fold_scalar_stats: double-precision exp() and 14 doubles in flight would otherwise set the register
// count of every launch that CAN reach them -- ppo_reduce_adam_kernel went from 85 to 156 registers that way, see there)
__device__ __noinline__ void fold_logstd_stats_raw(float raw, int n_act, int lane, double* __restrict__ info) {
  const bool has = lane < n_act;
  const double x = has ? fmin(fmax((double)raw, -20.0), 2.0) : 0.0;
  const double e = has ? exp(x) : 0.0;
  const double sm = wave_sum(x), sq = wave_sum(x * x), es = wave_sum(e), eq = wave_sum(e * e);
  const double mx = wave_max(has ? x : -INFINITY), mn = -wave_max(has ? -x : -INFINITY);
  const double emx = wave_max(has ? e : -INFINITY), emn = -wave_max(has ? -e : -INFINITY);
  if (lane == 0) {
    const double mean = sm / n_act, em = es / n_act;
    info[8] = mean;
    info[9] = n_act > 1 ? sqrt(fmax((sq - sm * mean) / (n_act - 1), 0.0)) : NAN;
    info[10] = mx; info[11] = mn;
    info[16] = em;
    info[17] = n_act > 1 ? sqrt(fmax((eq - es * em) / (n_act - 1), 0.0)) : NAN;
    info[18] = emx; info[19] = emn;
  }
}
__device__ __forceinline__ void fold_logstd_stats(const float* __restrict__ logstd, int n_act, int lane, double* __restrict__ info) {
  fold_logstd_stats_raw(logstd[lane < n_act ? lane : 0], n_act, lane, info);
}

// returns (wave 0 lanes) the reduced gradient value of chunk `bx` (64 parameters) of network `net`, 0 outside the parameter
// range; `stats`: this call also takes the network's scalar statistics (waves 2 / 3)
__device__ __forceinline__ float ppo_reduce_block(const float* __restrict__ partial,
                                                  const double* __restrict__ scal, int n_wg, int n_pf,
                                                  int p_stride, int p_pf, int p_vf,
                                                  const float* __restrict__ logstd, int n_act,
                                                  float* __restrict__ grads, double* __restrict__ info,
                                                  int net, int bx, bool stats) {
  // block = 64 consecutive parameters x RED_WAVES waves; wave w folds partials w, w + RED_WAVES, ... with RED_DEPTH
  // independent accumulators (fixed order => deterministic), then the waves fold through LDS.
  __shared__ float s_acc[RED_WAVES][RED_CHUNK];
  const int row0 = net == 0 ? 0 : n_pf, nrow = net == 0 ? n_pf : n_wg - n_pf;   // this network's partial rows
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = bx * RED_CHUNK + lane;
  const int pn = net == 0 ? p_pf : p_vf;
  s_acc[wave][lane] = fold_rows_wave(partial + (size_t)row0 * p_stride + p, nrow, p_stride, wave, p < pn);
  __syncthreads();
  float gval = 0.0f;
  if (wave == 0 && p < pn) {
    gval = fold_waves_tree(s_acc, lane);
    grads[(net == 0 ? 0 : p_pf) + p] = gval;
  }
  // Scalar statistics.  They used to sit in block (0, 0) -- three dependent rounds of loads over the workgroup partials and
  // then the log_std / std statistics as a serial double-precision loop (six exp() in ONE lane): ~5 us that the whole
  // launch waited for, twice the time of the fold itself.  Now the LAST chunk of each network's row takes them, off the
  // path of the blocks whose 64 parameters matter: wave 2 its network's statistics, wave 3 of the policy's block log_std
  // and std.
  if (stats && wave == 2) fold_scalar_stats(scal + (size_t)row0 * 8, nrow, net, lane, info);
  if (stats && net == 0 && wave == 3 && logstd) fold_logstd_stats(logstd, n_act, lane, info);
  return gval;
}

__global__ __launch_bounds__(64 * RED_WAVES) void ppo_reduce_kernel(const float* __restrict__ partial,
                                                         const double* __restrict__ scal, int n_wg, int n_pf,
                                                         int p_stride, int p_pf, int p_vf,
                                                         const float* __restrict__ logstd, int n_act,
                                                         float* __restrict__ grads, double* __restrict__ info) {
  ppo_reduce_block(partial, scal, n_wg, n_pf, p_stride, p_pf, p_vf, logstd, n_act, grads, info, blockIdx.y, blockIdx.x,
                   blockIdx.x == gridDim.x - 1);
}

// ---------------------------------------------------------------- K11 clip + Adam
// Every block recomputes the (tiny) group norms itself, so one launch does
// clip_grad_norm_ (coef = max_norm / (norm + 1e-6), clamped to 1) and Adam.
struct AdamDev {
  float* params; const float* grads; float* m; float* v;
  int n_groups; int off[5]; float lr[4];
  float max_norm, beta1, beta2, eps, grad_scale, bc1, bc2_sqrt;
  float* norms_out;
  double* step_state;         // {steps taken, beta1^steps, beta2^steps, -} or null (then bc1 / bc2_sqrt)
  const float* device_lr;     // per-group learning rates on the device, or null (then lr[])
  int self_tick;              // small grids: the last block to have read step_state advances it (no tick launch)
};
__global__ __launch_bounds__(256) void clip_adam_kernel(AdamDev a) {
  __shared__ float s_part[4][4];
  __shared__ float s_coef[4];
  __shared__ float s_bc[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (a.step_state && tid == 0) {                              // device-resident step: nothing in the launch changes
    s_bc[0] = (float)(1.0 - a.step_state[1] * (double)a.beta1);   // between replays of a captured graph
    s_bc[1] = (float)sqrt(1.0 - a.step_state[2] * (double)a.beta2);
    if (a.self_tick) {
      // Every block counts itself in AFTER its reads (release); the block that completes the count has therefore
      // seen all reads done (acquire) and advances the state for the next launch.  The counter lives in the reserved
      // 4th double and is back at zero when the kernel ends.
      unsigned* cnt = reinterpret_cast<unsigned*>(a.step_state + 3);
      const unsigned before = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (before == gridDim.x - 1) {
        a.step_state[0] += 1.0; a.step_state[1] *= (double)a.beta1; a.step_state[2] *= (double)a.beta2;
        __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  const bool need_norm = a.max_norm > 0.0f;                    // no clipping requested: skip the norm pass
  for (int g = 0; g < a.n_groups; ++g) {
    if (!need_norm) { if (lane == 0) s_part[g][wave] = 0.0f; continue; }
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int e = a.off[g] + tid;
    for (; e + 7 * 256 < a.off[g + 1]; e += 8 * 256) {         // 8 loads in flight per lane and round
      float x[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] = a.grads[e + 256 * k];
      s0 = fmaf(x[0], x[0], s0); s1 = fmaf(x[1], x[1], s1); s2 = fmaf(x[2], x[2], s2); s3 = fmaf(x[3], x[3], s3);
      s0 = fmaf(x[4], x[4], s0); s1 = fmaf(x[5], x[5], s1); s2 = fmaf(x[6], x[6], s2); s3 = fmaf(x[7], x[7], s3);
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
  const float bc1 = a.step_state ? s_bc[0] : a.bc1, bc2_sqrt = a.step_state ? s_bc[1] : a.bc2_sqrt;
  const int e = blockIdx.x * 256 + tid;
  if (e < a.off[a.n_groups]) {
    int g = 0;
    while (e >= a.off[g + 1]) ++g;
    const float gr = a.grads[e] * a.grad_scale * s_coef[g];
    const float m = a.beta1 * a.m[e] + (1.0f - a.beta1) * gr;
    const float v = a.beta2 * a.v[e] + (1.0f - a.beta2) * gr * gr;
    a.m[e] = m; a.v[e] = v;
    const float denom = sqrtf(v) / bc2_sqrt + a.eps;
    const float lr = a.device_lr ? a.device_lr[g] : a.lr[g];
    a.params[e] -= (lr / bc1) * (m / denom);
  }
}

// advances the device-resident step state after clip_adam_kernel has read it (stream order: every block is done)
__global__ void adam_tick_kernel(double* __restrict__ st, float beta1, float beta2) {
  st[0] += 1.0; st[1] *= (double)beta1; st[2] *= (double)beta2;
}

// Adam update of element e with the clip coefficient of its group
__device__ __forceinline__ void adam_element(const AdamDev& a, int e, float gr) {
  const float m = a.beta1 * a.m[e] + (1.0f - a.beta1) * gr;
  const float v = a.beta2 * a.v[e] + (1.0f - a.beta2) * gr * gr;
  a.m[e] = m; a.v[e] = v;
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  int g = 0;
  while (e >= a.off[g + 1]) ++g;
  a.params[e] -= (a.lr[g] / a.bc1) * (m / denom);
}

// Single-GPU path: partial reduce + clip_grad_norm_ + Adam in ONE launch.  The 2 x nb jobs (job j = the 64 parameters
// `j % nb` of network `j / nb`) are dealt to the blocks of a 1-D grid (block b: jobs b, b + grid, ...; the default grid is
// one block per job).  A block folds its jobs' parameters and publishes each one's sum of squares as ONE 64-bit
// agent-scope store {ss, epoch}; every block then polls the 2 x nb slots until all carry the epoch of this launch (the
// blocks are always co-resident: at most 180 small blocks on 256 CUs), derives the two group norms from them in the same
// fixed order (deterministic, identical in all blocks) and takes the Adam step for its own parameters -- straight from
// registers when it has one job.  No ticket, no reset: a slot is valid iff its epoch matches (epoch = Adam step count > 0;
// the workspace starts zeroed).  The poll is capped: a scheduling accident trips ws[0] instead of hanging.
// The GRID is the launch's resident footprint while it waits (for other ranks' gradients, with env shards on several
// ranks): blocks x 8 waves that stay on the device until every rank has delivered.  One rank per GPU: irrelevant.  Ranks
// sharing a device (tests): it must leave room for the other ranks' gradient kernels (trl_comm_set_wait_footprint).
// (at most 128 registers per wave: a block is 2 waves per SIMD, and two ranks sharing a device need TWO waiting launches'
// blocks -- 4 waves per SIMD -- resident beside each other.  With the statistics inlined into the job loop the kernel took
// 156 registers: the second rank's blocks found no room next to the first's, its norm rendezvous could not complete, and
// both ranks sat out their time-outs -- round 6, caught by tests/test_bench_multirank_gpu.py)
template <bool LOOP>                                  // false: the grid has one block per job (no job loop: the common launch)
__global__ __launch_bounds__(64 * RED_WAVES, 4) void ppo_reduce_adam_kernel(const float* __restrict__ partial,
                                                              const double* __restrict__ scal, int n_wg, int n_pf,
                                                              int p_stride, int p_pf, int p_vf,
                                                              const float* __restrict__ logstd, int n_act,
                                                              float* __restrict__ grads, double* __restrict__ info,
                                                              AdamDev a, float* __restrict__ ws, unsigned epoch,
                                                              int device_state, int xrank, XrArgs xr, int only_net) {
  __shared__ float s_coef[2];
  __shared__ float s_hyper[4];                                    // bc1, bc2_sqrt, lr_pf, lr_vf
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // only_net >= 0: the rows are ONE network's (a single-network gradient launch); its nb jobs only, its norm, its group
  const int nb = (p_stride + RED_CHUNK - 1) / RED_CHUNK, grid = gridDim.x, job0 = only_net == 1 ? nb : 0;
  const int n_jobs = only_net >= 0 ? job0 + nb : 2 * nb, blk = job0 + (int)blockIdx.x;
  // Graph-replayable form: the Adam step count and the learning rates live in the workspace header
  // (ws[1] = steps taken so far as uint32, ws[2..3] = lr), so no launch argument changes between replays.
  // Every block reads the count BEFORE it publishes its slots; block 0 bumps it only after it has seen
  // all slots, i.e. after every block has read it.
  // (the count was written by the previous launch, so a plain uniform load -- a scalar load that flies under the
  // reduction's vector loads -- is enough; the bias corrections are computed by an otherwise idle wave)
  double* bpow = reinterpret_cast<double*>(ws + 4);              // beta1^steps, beta2^steps of the steps taken so far
  double b1p = 0.0, b2p = 0.0;
  float lr0 = 0.0f, lr1 = 0.0f;
  if (device_state) {
    epoch = reinterpret_cast<const unsigned*>(ws)[1] + 1u;
    if (tid == 64 * (RED_WAVES - 1)) { b1p = bpow[0]; b2p = bpow[1]; lr0 = ws[2]; lr1 = ws[3]; }   // issued now, used after the fold
  }
  unsigned long long* slots = reinterpret_cast<unsigned long long*>(ws + 16);    // [2 nets][nb] {ss bits, epoch}
  // env shards on several ranks: the epoch of the cross-rank exchange is the communicator's own count of completed
  // gradient exchanges (read by every block before it publishes anything, advanced by block 0 at the end)
  // (two counts, ctl[4] and ctl[6]: a single-network launch of the value function -- the second of two concurrent update
  // chains -- counts its exchanges in ctl[6] and touches only the value function's granules, one of the policy in ctl[4];
  // a joint launch, which touches all granules, takes the larger of the two and leaves both at its epoch: a granule's stamps
  // only ever grow, whichever route wrote them last)
  unsigned xepoch = 0u;
  if (xrank) {
    const unsigned e4 = __hip_atomic_load(xr.ctl + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned e6 = __hip_atomic_load(xr.ctl + 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    xepoch = (only_net == 0 ? e4 : only_net == 1 ? e6 : (e4 > e6 ? e4 : e6)) + 1u;
  }
  float gval0 = 0.0f;                                             // wave 0: the (summed) gradient of this block's FIRST job
  for (int j = blk; j < (LOOP ? n_jobs : blk + 1); j += LOOP ? grid : 1) {
    const int net = j / nb, bx = j - net * nb;
    if (j != blk) __syncthreads();                                // the fold's LDS image is read by wave 0 of the previous job
    float gval = ppo_reduce_block(partial, scal, n_wg, n_pf, p_stride, p_pf, p_vf, logstd, n_act, grads, info, net, bx, bx == nb - 1);
    if (wave == 0) {
      if (xrank) {
        // C1 of SURVEY.md 8(e) inside the launch: every rank pushes its 64 folded values into its slot on all ranks and
        // sums the slots in rank order (trl_comm.h) -- the gradient every rank clips and steps with is the same, bit for bit
        const int pe = bx * RED_CHUNK + lane;
        const bool act = pe < (net == 0 ? p_pf : p_vf);
        const int ge = (net == 0 ? 0 : p_pf) + pe;
        gval = xr_allsum_f32(xr, xepoch, act ? ge : 0, gval, act);
        if (act) grads[ge] = gval;
      }
      if (j == blk) gval0 = gval;
      const float ss = wave_sum(gval * gval);
      if (lane == 0)
        __hip_atomic_store(slots + net * nb + bx, ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(ss),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (device_state && tid == 64 * (RED_WAVES - 1)) {
    b1p *= (double)a.beta1; b2p *= (double)a.beta2;
    s_hyper[0] = (float)(1.0 - b1p);
    s_hyper[1] = (float)sqrt(1.0 - b2p);
    s_hyper[2] = lr0; s_hyper[3] = lr1;
  }
  if (LOOP && blk >= n_jobs) return;                              // (a grid larger than the job list)
  // the optimiser state of this block's first job is requested now and arrives while the norm slots are polled
  const int net_ = blk / nb, pe_ = (blk - net_ * nb) * RED_CHUNK + lane;
  const bool own_ = wave == 0 && pe_ < (net_ == 0 ? p_pf : p_vf);
  const int ge_ = (net_ == 0 ? 0 : p_pf) + pe_;
  float m_old = 0.0f, v_old = 0.0f, p_old = 0.0f;
  if (own_) { m_old = a.m[ge_]; v_old = a.v[ge_]; p_old = a.params[ge_]; }
  // ---- group norms (pf, vf): wave w polls net w's slots, then sums them in fixed order ----
  if (wave < 2 && (only_net < 0 || wave == only_net)) {
    float acc = 0.0f;
    for (int b = lane; b < nb; b += 64) {
      unsigned long long v;
      unsigned it = 0;
      unsigned long long t0 = 0;
      while ((unsigned)((v = __hip_atomic_load(slots + wave * nb + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != epoch) {
        __builtin_amdgcn_s_sleep(1);
        if ((++it & 1023u) == 0) {                                // bounded by wall-clock time (100 MHz counter): ~2 s alone,
          const unsigned long long now = wall_clock64();          // ~25 s when other ranks' gradients are being waited for
          if (t0 == 0) t0 = now;
          if (now - t0 > (xrank ? xr.wait_ticks + xr.wait_ticks / 4 : 200000000ull)) {
            __hip_atomic_store(reinterpret_cast<unsigned*>(ws), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
      }
      acc += __uint_as_float((unsigned)v);
    }
    acc = wave_sum(acc) * a.grad_scale * a.grad_scale;
    if (lane == 0) {
      const float norm = sqrtf(acc);
      s_coef[wave] = (a.max_norm > 0.0f) ? fminf(a.max_norm / (norm + 1e-6f), 1.0f) : 1.0f;
      if (a.norms_out && blk == job0) a.norms_out[wave] = norm;
    }
  }
  __syncthreads();
  // (locals, not fields of `a`: writing into the by-value argument struct moves it to scratch memory and turns every access
  // through its pointers into a flat instruction)
  float bc1 = a.bc1, bc2_sqrt = a.bc2_sqrt, lr_pf = a.lr[0], lr_vf = a.lr[1];
  if (device_state) {
    bc1 = s_hyper[0]; bc2_sqrt = s_hyper[1]; lr_pf = s_hyper[2]; lr_vf = s_hyper[3];
    if (blk == job0) {                                           // every block has read the header by now (see above)
      if (tid == 0) __hip_atomic_store(reinterpret_cast<unsigned*>(ws) + 1, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tid == 64 * (RED_WAVES - 1)) { bpow[0] = b1p; bpow[1] = b2p; }
    }
  }
  // (every block has passed its exchanges by the time block 0 has seen all norm slots)
  if (xrank && blk == job0 && tid == 0) {
    if (only_net != 1) __hip_atomic_store(xr.ctl + 4, xepoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (only_net != 0) __hip_atomic_store(xr.ctl + 6, xepoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // A cross-rank wait that timed out left a PARTIAL gradient sum: no parameter is written then (every block has finished
  // its exchanges before any block sees all norm slots, so the flag is final here and all blocks decide alike); the host
  // raises through trl_comm_error at its next check.
  const bool xfail = xrank && __hip_atomic_load(xr.ctl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
  if (wave == 0 && !xfail) {                                       // adam_element, the first job on the prefetched state
    for (int j = blk; j < (LOOP ? n_jobs : blk + 1); j += LOOP ? grid : 1) {
      const int net = j / nb, pe = (j - net * nb) * RED_CHUNK + lane;
      if (pe >= (net == 0 ? p_pf : p_vf)) continue;
      const int ge = (net == 0 ? 0 : p_pf) + pe;
      const bool first = j == blk;
      const float g_ = first ? gval0 : grads[ge];                  // (later jobs: this lane's own store of the fold above)
      const float mo = first ? m_old : a.m[ge], vo = first ? v_old : a.v[ge], po = first ? p_old : a.params[ge];
      const float gr = g_ * a.grad_scale * s_coef[net];
      const float m = a.beta1 * mo + (1.0f - a.beta1) * gr;
      const float v = a.beta2 * vo + (1.0f - a.beta2) * gr * gr;
      a.m[ge] = m; a.v[ge] = v;
      const float denom = sqrtf(v) / bc2_sqrt + a.eps;
      a.params[ge] = po - ((net == 0 ? lr_pf : lr_vf) / bc1) * (m / denom);
    }
  }
}

// ---------------------------------------------------------------- the whole minibatch step in ONE launch
// trl_ppo_minibatch_step_f32: the gradient pass above, then -- inside the same launch -- what ppo_reduce_adam_kernel does
// (fold of the workgroups' partial rows, clip_grad_norm_, Adam; ppo.py:72-74, 117-122).  Every workgroup of the gradient
// grid is resident at once (one per CU: 416 registers per wave, 100 KB of LDS), so they can meet:
//   1. a workgroup stores its partial row and scalar row, makes them visible (agent-scope release) and raises its arrival
//      flag = this launch's sequence number;
//   2. it polls all n_wg flags (one per thread), then (acquire) owns fold jobs c = wg, wg + n_wg, ...: job c < 2 nb is the
//      64 parameters `c % nb` of network `c / nb` -- the fold of ppo_reduce_block, same association order, so the step is
//      bit-identical to the two-launch sequence -- jobs 2 nb .. 2 nb + 2 are the scalar statistics;
//   3. it publishes its chunk's sum of squares as a {value, sequence} granule, polls the 2 nb granules, derives both clip
//      coefficients in the fixed order and steps its own 64 parameters.
// No ticket, no reset: flags and granules are valid iff they carry this launch's sequence number, which the kernel keeps in
// its workspace (read by every workgroup before it raises its flag, advanced by workgroup 0 after it has seen every granule).
// A wait that does not complete within ~2 s (the grid was not co-resident: something else held CUs for that long) sets
// workspace word 0 and info[23] and leaves the parameters untouched.
// Saves the dependent launch (~6 us of latency floor, measured) of every one of the 40 updates of an iteration.
struct StepDev {
  float* grads; double* info; float* ws; unsigned epoch; int device_state;
  const float* logstd; int n_act, p_pf, p_vf;
  AdamDev adam;
};
#define STEP_FLAGS 256                              /* arrival flags = threads of a workgroup */
__host__ __device__ inline int step_ws_off(int nb) { return 16 + 4 * nb; }          // behind ppo_reduce_adam_kernel's region
__host__ __device__ inline int step_ws_words(int nb) { return step_ws_off(nb) + 4 + STEP_FLAGS + 4 * nb; }

// STEP_CLK (tools/ab_step.py): phase stamps of every workgroup behind the workspace.  Transfers that were built and measured
// on MI355X before this one (profiles/NOTES_r06.md): plain row stores + agent-scope release / acquire fences (every wave's
// release walks its XCD's L2: +21 us per launch), {value, sequence} granules polled by the fold itself (shortest tail, but
// twice the bytes through the coherence point: the pass ends 3.7 us later).
#ifdef STEP_CLK
#define SCLK(k) { if (tid == 0) clk[k] = wall_clock64(); }
#else
#define SCLK(k)
#endif

// all n_wg arrival flags carry `seq` (block-uniform result; false: timed out).  Wave 0 polls, four flags per lane.
__device__ __forceinline__ bool step_wait_flags(const unsigned* flags, int n_wg, unsigned seq, int* s_fail) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave == 0) {
    unsigned long long t0 = 0;
    for (unsigned it = 1;; ++it) {
      bool ok = true;
#pragma unroll
      for (int q = 0; q < STEP_FLAGS / 64; ++q) {
        const int w = lane + 64 * q;
        const unsigned f = __hip_atomic_load(flags + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = ok && (w >= n_wg || f == seq);
      }
      if (__all(ok)) break;
      if ((it & 63u) == 0) {
        const unsigned long long now = wall_clock64();          // 100 MHz
        if (t0 == 0) t0 = now;
        else if (now - t0 > 200000000ull) { if (lane == 0) *s_fail = 1; break; }
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  return *s_fail == 0;
}

__device__ __forceinline__ void ppo_step_tail(const PpoDev& a, const StepDev& t, float* lds, unsigned seq) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x, n_wg = a.n_wg;
  const int nb = (a.p_stride + RED_CHUNK - 1) / RED_CHUNK;
  unsigned* wsu = reinterpret_cast<unsigned*>(t.ws);
  unsigned* seqp = wsu + step_ws_off(nb);
  unsigned* flags = seqp + 4;
  unsigned long long* slots = reinterpret_cast<unsigned long long*>(flags + STEP_FLAGS);   // [2 nets][nb] {ss bits, seq}
#ifdef STEP_CLK
  unsigned long long* clk = reinterpret_cast<unsigned long long*>(wsu + step_ws_words(nb)) + (size_t)wg * 8;
#endif
  float (*s_acc)[RED_CHUNK] = reinterpret_cast<float (*)[RED_CHUNK]>(lds);
  float* s_coef = lds + RED_WAVES * RED_CHUNK;                   // [2]
  float* s_hyper = s_coef + 2;                                   // bc1, bc2_sqrt, lr_pf, lr_vf
  int* s_fail = reinterpret_cast<int*>(s_hyper + 4);
  SCLK(1)
  // ---- header: read before the arrival flag goes up (workgroup 0 advances it after the last rendezvous) ----
  unsigned epoch = t.epoch;
  double* bpow = reinterpret_cast<double*>(t.ws + 4);
  double b1p = 0.0, b2p = 0.0;
  float lr0 = 0.0f, lr1 = 0.0f;
  if (t.device_state) {
    epoch = wsu[1] + 1u;
    if (tid == WV_THREADS - 1) { b1p = bpow[0]; b2p = bpow[1]; lr0 = t.ws[2]; lr1 = t.ws[3]; }
  }
  // log_std as it is BEFORE this step (its statistics are the pre-step ones, ppo.py:82-85): workgroup 0 reads it here --
  // the load has returned by the barriers below, i.e. before this workgroup publishes the norm granule of its chunk, and
  // no workgroup steps a parameter before it has seen every granule
  float ls_raw = 0.0f;
  if (wg == 0 && wave == WV_WAVES - 1 && t.logstd && lane < t.n_act) ls_raw = t.logstd[lane];
  // ---- 1. this workgroup's rows are out: arrive ----
  // the rows went out as device-scope write-through stores: they are at the coherence point once the wave's store counter
  // has drained -- no agent-scope release fence (an L2 write-back walk per wave)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);
  if (tid == 0) *s_fail = 0;
  __syncthreads();                                               // (also: the pass's LDS is dead from here)
  if (tid == 0) __hip_atomic_store(flags + wg, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  SCLK(2)
  // ---- 2. every workgroup's rows ----
  if (!step_wait_flags(flags, n_wg, seq, s_fail)) {
    if (tid == 0) { __hip_atomic_store(wsu, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); t.info[23] = 1.0; }
    return;
  }
  SCLK(3)
  SCLK(4)
  if (t.device_state && tid == WV_THREADS - 1) {
    b1p *= (double)t.adam.beta1; b2p *= (double)t.adam.beta2;
    s_hyper[0] = (float)(1.0 - b1p);
    s_hyper[1] = (float)sqrt(1.0 - b2p);
    s_hyper[2] = lr0; s_hyper[3] = lr1;
  }
  // ---- 3. fold jobs (rows: device-coherent loads, not served from this XCD's L2) ----
  const int n_jobs = 2 * nb + 2;                                 // the chunks of both networks, then their scalar statistics
  float gval0 = 0.0f;                                            // wave 0: the folded gradient of this workgroup's FIRST chunk
  bool has_chunk = false;
  for (int c = wg; c < n_jobs; c += n_wg) {
    if (c < 2 * nb) {
      const int net = c / nb, bx = c - net * nb;
      const int row0 = net == 0 ? 0 : a.n_pf, nrow = net == 0 ? a.n_pf : n_wg - a.n_pf;
      const int p = bx * RED_CHUNK + lane, pn = net == 0 ? t.p_pf : t.p_vf;
      if (has_chunk) __syncthreads();                            // s_acc is read by wave 0 of the previous job
      float r0, r1;
      fold_rows_wave_pair<true>(a.partial + (size_t)row0 * a.p_stride + p, nrow, a.p_stride, wave, p < pn, r0, r1);
      s_acc[wave][lane] = r0; s_acc[wave + 4][lane] = r1;
      __syncthreads();
      if (wave == 0) {
        float gval = 0.0f;
        if (p < pn) { gval = fold_waves_tree(s_acc, lane); t.grads[(net == 0 ? 0 : t.p_pf) + p] = gval; }
        if (!has_chunk) gval0 = gval;
        const float ss = wave_sum(gval * gval);
        if (lane == 0)
          __hip_atomic_store(slots + net * nb + bx, ((unsigned long long)seq << 32) | (unsigned long long)__float_as_uint(ss),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      has_chunk = true;
    } else if (wave == 1) {                                      // a network's loss / log-prob / value statistics
      const int j = c - 2 * nb;
      fold_scalar_stats<true>(a.scal_partial + (size_t)(j == 0 ? 0 : a.n_pf) * 8, j == 0 ? a.n_pf : n_wg - a.n_pf, j, lane, t.info);
    }
  }
  if (wg == 0 && wave == WV_WAVES - 1 && t.logstd) fold_logstd_stats_raw(ls_raw, t.n_act, lane, t.info);
  SCLK(5)
  if (!has_chunk) { SCLK(6) SCLK(7) return; }                     // (workgroup 0 always has one)
  // the optimiser state of the first chunk is requested now and arrives while the norm granules are polled
  const int net0 = wg / nb, pe0 = (wg - net0 * nb) * RED_CHUNK + lane;
  const bool own0 = wave == 0 && pe0 < (net0 == 0 ? t.p_pf : t.p_vf);
  const int ge0 = (net0 == 0 ? 0 : t.p_pf) + pe0;
  float m_old = 0.0f, v_old = 0.0f, p_old = 0.0f;
  if (own0) { m_old = t.adam.m[ge0]; v_old = t.adam.v[ge0]; p_old = t.adam.params[ge0]; }
  // ---- 4. group norms (pf, vf): wave w polls net w's granules, then sums them in fixed order ----
  if (wave < 2) {
    float acc = 0.0f;
    bool fail = false;
    for (int b = lane; b < nb; b += 64) {
      unsigned long long v;
      unsigned it = 0;
      unsigned long long t0 = 0;
      while ((unsigned)((v = __hip_atomic_load(slots + wave * nb + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != seq) {
        __builtin_amdgcn_s_sleep(1);
        if ((++it & 1023u) == 0) {
          const unsigned long long now = wall_clock64();
          if (t0 == 0) t0 = now;
          if (now - t0 > 200000000ull) { fail = true; break; }
        }
      }
      acc += __uint_as_float((unsigned)v);
    }
    if (fail) *s_fail = 1;
    acc = wave_sum(acc) * t.adam.grad_scale * t.adam.grad_scale;
    if (lane == 0) {
      const float norm = sqrtf(acc);
      s_coef[wave] = (t.adam.max_norm > 0.0f) ? fminf(t.adam.max_norm / (norm + 1e-6f), 1.0f) : 1.0f;
      if (t.adam.norms_out && wg == 0) t.adam.norms_out[wave] = norm;
    }
  }
  __syncthreads();
  SCLK(6)
  if (*s_fail) {
    if (tid == 0) { __hip_atomic_store(wsu, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); t.info[23] = 1.0; }
    return;
  }
  float bc1 = t.adam.bc1, bc2_sqrt = t.adam.bc2_sqrt, lr_pf = t.adam.lr[0], lr_vf = t.adam.lr[1];
  if (t.device_state) { bc1 = s_hyper[0]; bc2_sqrt = s_hyper[1]; lr_pf = s_hyper[2]; lr_vf = s_hyper[3]; }
  if (wg == 0) {                                                 // every workgroup has read the header by now (see above)
    if (tid == 0) {
      __hip_atomic_store(seqp, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t.device_state) __hip_atomic_store(wsu + 1, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (t.device_state && tid == WV_THREADS - 1) { bpow[0] = b1p; bpow[1] = b2p; }
  }
  // ---- 5. Adam on the own chunks ----
  if (wave == 0) {
    bool first = true;
    for (int c = wg; c < 2 * nb; c += n_wg, first = false) {
      const int net = c / nb, pe = (c - net * nb) * RED_CHUNK + lane;
      if (pe >= (net == 0 ? t.p_pf : t.p_vf)) continue;
      const int ge = (net == 0 ? 0 : t.p_pf) + pe;
      const float g_ = first ? gval0 : t.grads[ge];                 // (later chunks: this lane's own store of step 3)
      const float mo = first ? m_old : t.adam.m[ge], vo = first ? v_old : t.adam.v[ge], po = first ? p_old : t.adam.params[ge];
      const float gr = g_ * t.adam.grad_scale * s_coef[net];
      const float m = t.adam.beta1 * mo + (1.0f - t.adam.beta1) * gr;
      const float v = t.adam.beta2 * vo + (1.0f - t.adam.beta2) * gr * gr;
      t.adam.m[ge] = m; t.adam.v[ge] = v;
      const float denom = sqrtf(v) / bc2_sqrt + t.adam.eps;
      t.adam.params[ge] = po - ((net == 0 ? lr_pf : lr_vf) / bc1) * (m / denom);
    }
  }
  SCLK(7)
}

template <int D, int H, int A, int ACT, bool CONTIG, bool RT, bool STEP>
__global__ __launch_bounds__(WV_THREADS, 1) void ppo_grad_wave_kernel(PpoDev a, StepDev t) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // this launch's sequence number (the step's workspace): read by every workgroup before anything of it is published
  unsigned seq = 0u;
  if constexpr (STEP)
    seq = __hip_atomic_load(reinterpret_cast<unsigned*>(t.ws) + step_ws_off((a.p_stride + RED_CHUNK - 1) / RED_CHUNK),
                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
#ifdef STEP_CLK
  if (STEP && threadIdx.x == 0)
    (reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned*>(t.ws) + step_ws_words((a.p_stride + RED_CHUNK - 1) / RED_CHUNK)) + (size_t)blockIdx.x * 8)[0] = wall_clock64();
#endif
#ifdef TRL_CHAIN_CLK                                  /* tools/time_chains.py: when did this workgroup start and end (100 MHz) */
  const unsigned long long clk0 = wall_clock64();
#endif
  if ((int)blockIdx.x < a.n_pf) ppo_wave_pass<D, H, A, ACT, true, CONTIG, RT, STEP>(a, lds, blockIdx.x, a.n_pf);
  else                          ppo_wave_pass<D, H, A, ACT, false, CONTIG, RT, STEP>(a, lds, blockIdx.x - a.n_pf, a.n_wg - a.n_pf);
#ifdef TRL_CHAIN_CLK
  __syncthreads();
  if (threadIdx.x == 0) {                             // the row's unused 8th scalar and the last three padding floats of the partial row
    a.scal_partial[(size_t)blockIdx.x * 8 + 7] = (double)wall_clock64();
    float* pad = a.partial + (size_t)blockIdx.x * a.p_stride + a.p_stride - 3;
    pad[0] = (float)(clk0 >> 40); pad[1] = (float)((clk0 >> 20) & 0xFFFFFull); pad[2] = (float)(clk0 & 0xFFFFFull);
  }
#endif
  if constexpr (STEP) ppo_step_tail(a, t, lds, seq);
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

// Shapes the fused minibatch kernels carry: the benchmark shape as a compile-time instantiation, and any D in [2, 32],
// A in [1, 8] at H = 64 through the runtime-dims instantiations (ppo_wave_pass<..., RT = true>: the 17- and the 32-wide tile).
static bool ppo_shape_rt(int D, int H, int A) { return H == 64 && D >= 2 && D <= 32 && A >= 1 && A <= 8; }
extern "C" int trl_ppo_partial_stride(int D, int H, int A) {
  if (SHAPE_IS(17, 64, 6)) return PpoShape<17, 64, 6>::P_STRIDE;
  if (ppo_shape_rt(D, H, A)) {
    const int p_pf = H * D + H + H * H + H + A * H + A + A, p_vf = H * D + H + H * H + H + H + 1;
    return ((p_pf > p_vf ? p_pf : p_vf) + 63) & ~63;
  }
  trl_set_error("trl_ppo_partial_stride: shape D=%d H=%d A=%d not instantiated", D, H, A);
  return TRL_EUNSUPPORTED;
}

// Kernel generations that were built and measured on MI355X before this one (profiles/README.md): groups of
// 4 waves per tile with LDS rendezvous (85 us per 65 536-sample minibatch), this wave-per-tile kernel (64 us),
// and a PAIR of waves per tile with two waves per SIMD (75 us) -- on gfx950 the fp32 MFMA and the VALU do not
// overlap across the two waves of a SIMD (tools/ubench/mfma_valu.hip: an MFMA-only wave and a VALU-only wave
// on one SIMD take the SUM of their times), so a second wave only adds its duplicated loss / fetch work.
template <int D, int H, int A, int ACT, bool CONTIG, bool RT, bool STEP>
static int launch_ppo_v(const PpoDev& d, const StepDev& t, hipStream_t s) {
  using S = WvShape<D, H, A>;
  const size_t lds = S::LDS_FLOATS * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)ppo_grad_wave_kernel<D, H, A, ACT, CONTIG, RT, STEP>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { trl_set_error("ppo_grad: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  hipLaunchKernelGGL((ppo_grad_wave_kernel<D, H, A, ACT, CONTIG, RT, STEP>), dim3(d.n_wg), dim3(WV_THREADS), lds, s, d, t);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}
// N % 16 == 0 (every tile = 16 consecutive envs of one time row): scalar tile addressing; any other N: per-lane cells.
// `t` non-null: the whole step (fold, clip, Adam) in the same launch.
template <int D, int H, int A, int ACT, bool RT>
static int launch_ppo(const PpoDev& d, const StepDev* t, hipStream_t s) {
  static const StepDev none{};
  if (t) return (d.N % 16 == 0) ? launch_ppo_v<D, H, A, ACT, true, RT, true>(d, *t, s) : launch_ppo_v<D, H, A, ACT, false, RT, true>(d, *t, s);
  return (d.N % 16 == 0) ? launch_ppo_v<D, H, A, ACT, true, RT, false>(d, none, s) : launch_ppo_v<D, H, A, ACT, false, RT, false>(d, none, s);
}

// Policy / value split of the grid.  A policy tile costs more than a value tile (head, log-prob loss,
// dW3 on the matrix pipe: 16.7 k vs 12.1 k cycles measured at D=17, H=64, A=6), and every wave runs whole
// tiles, so the split that minimises the slower side's ceil(tiles / waves) * cost is searched directly.
extern "C" int trl_ppo_wg_split(int D, int H, int A, int n_tiles, int n_wg) {
  if (n_wg < 2 || n_tiles <= 0) { trl_set_error("trl_ppo_wg_split: need n_wg >= 2 and n_tiles > 0"); return TRL_EINVAL; }
  const double c_pf = 16.7, c_vf = 12.1;
  int best = n_wg / 2;
  double best_t = 1e300;
  for (int x = 1; x < n_wg; ++x) {
    const double t_pf = (double)((n_tiles + WV_WAVES * x - 1) / (WV_WAVES * x)) * c_pf;
    const double t_vf = (double)((n_tiles + WV_WAVES * (n_wg - x) - 1) / (WV_WAVES * (n_wg - x))) * c_vf;
    const double t = t_pf > t_vf ? t_pf : t_vf;
    if (t < best_t - 1e-9 || (t < best_t + 1e-9 && abs(2 * x - n_wg) < abs(2 * best - n_wg))) { best_t = t; best = x; }
  }
  (void)D; (void)H; (void)A;
  return best;
}
// n_wg_pf: 0 = even split, [1, n_wg) = that many policy workgroups, n_wg = ALL workgroups run the policy, -1 = all run the
// value net (one network per launch: the two update chains of PPO are independent -- ppo.py:93-122 vs 41-91 -- and may run
// as separate, concurrent launch sequences)
static int resolve_pf_wgs(int n_wg, int n_wg_pf) { return n_wg_pf < 0 ? 0 : (n_wg_pf > 0 ? n_wg_pf : n_wg / 2); }

static int ppo_grad_launch(const trl_ppo_batch_t* p, const StepDev* step, void* stream) {
  if (!p) { trl_set_error("ppo_grad: null descriptor"); return TRL_EINVAL; }
  TRL_REQUIRE(p->obs && p->acts && p->advs && p->rets, "null rollout tensor");
  TRL_REQUIRE(p->loss_mode == TRL_LOSS_PPO_CLIP || p->loss_mode == TRL_LOSS_A2C, "unknown loss_mode");
  TRL_REQUIRE(p->loss_mode == TRL_LOSS_A2C || p->old_logp, "the clipped surrogate needs old_logp");
  TRL_REQUIRE(!p->clipped_value_loss || p->old_values, "the clipped value loss needs old_values");
  TRL_REQUIRE(p->adv_raw && (p->pf_params || p->vf_params) && p->partial && p->scal_partial, "null pointer");
  TRL_REQUIRE(p->rows_mb > 0 && p->N > 0, "empty minibatch");
  TRL_REQUIRE(p->n_wg >= 1 && (p->n_wg >= 2 || p->n_wg_pf == -1 || p->n_wg_pf == p->n_wg), "n_wg must be >= 2 (>= 1 for one network)");
  TRL_REQUIRE(p->n_wg_pf >= -1 && p->n_wg_pf <= p->n_wg, "n_wg_pf must be 0 (even split), in [1, n_wg], or -1 (value net only)");
  TRL_REQUIRE(p->n_wg_pf == p->n_wg || p->vf_params, "null value parameters");
  TRL_REQUIRE(p->n_wg_pf == -1 || p->pf_params, "null policy parameters");
  TRL_REQUIRE(p->n_global > 1.0, "n_global must exceed 1 (unbiased std)");
  const int D = p->D, H = p->H, A = p->A;
  TRL_REQUIRE(((uintptr_t)p->partial & 15) == 0 && (((uintptr_t)p->pf_params | (uintptr_t)p->vf_params) & 3) == 0,
              "partial rows must be 16-byte aligned, parameter blocks 4-byte aligned");
  TRL_REQUIRE(!SHAPE_IS(17, 64, 6) || (((uintptr_t)p->pf_params | (uintptr_t)p->vf_params) & 15) == 0,
              "parameter blocks of the benchmark shape must be 16-byte aligned");
  PpoDev d;
  d.obs = p->obs; d.acts = p->acts; d.advs = p->advs; d.rets = p->rets; d.old_values = p->old_values;
  d.old_logp = p->old_logp; d.row_idx = p->row_idx; d.rows_mb = p->rows_mb; d.N = p->N;
  d.adv_raw = p->adv_raw; d.n_global = p->n_global; d.pf_params = p->pf_params; d.vf_params = p->vf_params;
  d.clip_para = p->clip_para; d.entropy_coeff = p->entropy_coeff;
  d.clipped_value_loss = p->clipped_value_loss; d.tanh_action = p->tanh_action; d.loss_mode = p->loss_mode;
  d.partial = p->partial; d.scal_partial = p->scal_partial; d.n_wg = p->n_wg;
  d.n_pf = resolve_pf_wgs(p->n_wg, p->n_wg_pf);
  d.D = D; d.A = A;
  hipStream_t s = (hipStream_t)stream;
  if (SHAPE_IS(17, 64, 6)) {
    d.p_stride = PpoShape<17, 64, 6>::P_STRIDE;
    if (p->act == TRL_ACT_TANH) return launch_ppo<17, 64, 6, TRL_ACT_TANH, false>(d, step, s);
    if (p->act == TRL_ACT_RELU) return launch_ppo<17, 64, 6, TRL_ACT_RELU, false>(d, step, s);
  } else if (ppo_shape_rt(D, H, A) && D <= 17) {      // actual dims at run time inside the (17, 64, 8) tile
    d.p_stride = trl_ppo_partial_stride(D, H, A);
    if (p->act == TRL_ACT_TANH) return launch_ppo<17, 64, 8, TRL_ACT_TANH, true>(d, step, s);
    if (p->act == TRL_ACT_RELU) return launch_ppo<17, 64, 8, TRL_ACT_RELU, true>(d, step, s);
  } else if (ppo_shape_rt(D, H, A)) {                 // 18 .. 32 input features: the (32, 64, 8) tile
    d.p_stride = trl_ppo_partial_stride(D, H, A);
    if (p->act == TRL_ACT_TANH) return launch_ppo<32, 64, 8, TRL_ACT_TANH, true>(d, step, s);
    if (p->act == TRL_ACT_RELU) return launch_ppo<32, 64, 8, TRL_ACT_RELU, true>(d, step, s);
  }
  trl_set_error("ppo_grad: shape D=%d H=%d A=%d act=%d not instantiated", D, H, A, p->act);
  return TRL_EUNSUPPORTED;
}

extern "C" int trl_ppo_minibatch_grad_f32(const trl_ppo_batch_t* p, void* stream) { return ppo_grad_launch(p, nullptr, stream); }

extern "C" int trl_ppo_reduce_f32(const float* partial, const double* scal_partial, int n_wg, int n_wg_pf, int D,
                                  int H, int A, const float* pf_params, float* grads, double* info, void* stream) {
  TRL_REQUIRE(partial && scal_partial && grads && info, "null pointer");
  TRL_REQUIRE(n_wg >= 2 && n_wg_pf >= 0 && n_wg_pf < n_wg, "need n_wg >= 2 and n_wg_pf in [0, n_wg)");
  const int ps = trl_ppo_partial_stride(D, H, A);
  if (ps < 0) return ps;
  const int p_pf = H * D + H + H * H + H + A * H + A + A, p_vf = H * D + H + H * H + H + H + 1;
  hipLaunchKernelGGL(ppo_reduce_kernel, dim3(trl_ceil_div(ps, RED_CHUNK), 2), dim3(64 * RED_WAVES), 0, (hipStream_t)stream,
                     partial, scal_partial, n_wg, resolve_pf_wgs(n_wg, n_wg_pf), ps, p_pf, p_vf,
                     pf_params ? pf_params + (p_pf - A) : (const float*)nullptr, A, grads, info);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

static int fill_adam(const trl_adam_t* p, AdamDev& d) {
  if (!p) { trl_set_error("clip_adam: null descriptor"); return TRL_EINVAL; }
  TRL_REQUIRE(p->params && p->grads && p->exp_avg && p->exp_avg_sq, "null pointer");
  TRL_REQUIRE(p->n_groups >= 1 && p->n_groups <= 4, "n_groups must be 1..4");
  TRL_REQUIRE(p->step_count >= 1 || p->device_state || p->step_state, "step_count starts at 1");
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
  d.step_state = p->step_state;
  d.device_lr = p->device_lr;
  d.self_tick = 0;
  return TRL_OK;
}

extern "C" int trl_ppo_reduce_adam_workspace(int D, int H, int A) {
  const int ps = trl_ppo_partial_stride(D, H, A);
  if (ps < 0) return ps;
  return 16 + 4 * trl_ceil_div(ps, RED_CHUNK);       // header + {ss, epoch} per block and network
}

static int launch_reduce_adam(const float* partial, const double* scal_partial, int n_wg, int n_wg_pf, int D, int H, int A,
                              float* grads, double* info, const trl_adam_t* adam, float* workspace, const XrArgs* xr,
                              int max_blocks, void* stream, int only_net = -1) {
  TRL_REQUIRE(partial && scal_partial && grads && info && workspace, "null pointer");
  TRL_REQUIRE(only_net >= 0 ? n_wg >= 1 : (n_wg >= 2 && n_wg_pf >= 0 && n_wg_pf < n_wg), "need n_wg >= 2 and n_wg_pf in [0, n_wg)");
  const int ps = trl_ppo_partial_stride(D, H, A);
  if (ps < 0) return ps;
  AdamDev d;
  int rc = fill_adam(adam, d);
  if (rc) return rc;
  const int p_pf = H * D + H + H * H + H + A * H + A + A, p_vf = H * D + H + H * H + H + H + 1;
  TRL_REQUIRE(adam->n_groups == 2 && adam->group_sizes[0] == p_pf && adam->group_sizes[1] == p_vf,
              "optimiser groups must be [policy | value] of this shape");
  TRL_REQUIRE(adam->grads == grads, "adam->grads must be the reduce output");
  TRL_REQUIRE(!xr || p_pf + p_vf <= TRL_XR_CAP_GRAD, "gradient exceeds the peer buffer");
  XrArgs none;
  none.rank = 0; none.world = 1; none.ctl = nullptr; none.wait_ticks = 0;
  for (int r = 0; r < TRL_MAX_RANKS; ++r) none.peer[r] = nullptr;
  int grid = (only_net >= 0 ? 1 : 2) * trl_ceil_div(ps, RED_CHUNK);   // one block per job, unless the caller bounds the footprint
  if (max_blocks > 0 && max_blocks < grid) grid = max_blocks;
  const int n_pf = only_net == 0 ? n_wg : (only_net == 1 ? 0 : resolve_pf_wgs(n_wg, n_wg_pf));
  const int jobs = (only_net >= 0 ? 1 : 2) * trl_ceil_div(ps, RED_CHUNK);
  if (grid == jobs)
    hipLaunchKernelGGL(ppo_reduce_adam_kernel<false>, dim3(grid), dim3(64 * RED_WAVES), 0, (hipStream_t)stream,
                       partial, scal_partial, n_wg, n_pf, ps, p_pf, p_vf,
                       (const float*)(adam->params + (p_pf - A)), A, grads, info, d, workspace, (unsigned)adam->step_count,
                       adam->device_state, xr ? 1 : 0, xr ? *xr : none, only_net);
  else
    hipLaunchKernelGGL(ppo_reduce_adam_kernel<true>, dim3(grid), dim3(64 * RED_WAVES), 0, (hipStream_t)stream,
                       partial, scal_partial, n_wg, n_pf, ps, p_pf, p_vf,
                       (const float*)(adam->params + (p_pf - A)), A, grads, info, d, workspace, (unsigned)adam->step_count,
                       adam->device_state, xr ? 1 : 0, xr ? *xr : none, only_net);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

extern "C" int trl_ppo_reduce_adam_f32(const float* partial, const double* scal_partial, int n_wg, int n_wg_pf,
                                       int D, int H, int A, float* grads, double* info, const trl_adam_t* adam,
                                       float* workspace, void* stream) {
  return launch_reduce_adam(partial, scal_partial, n_wg, n_wg_pf, D, H, A, grads, info, adam, workspace, nullptr, 0, stream);
}

// ONE network's half of the step: `partial` / `scal_partial` hold the n_wg rows of a single-network gradient launch
// (trl_ppo_minibatch_grad_f32 with n_wg_pf = n_wg: net 0, the policy; n_wg_pf = -1: net 1, the value function); the fold,
// that group's clip and its Adam step.  PPO's critic and actor updates are independent (separate networks, optimisers
// and clips: ppo.py:93-122 / 41-91), so the two halves may run as separate launch sequences, concurrently, each with its
// own workspace (own Adam header: both count the same steps).  Same arithmetic and order as the joint launch: bit-identical.
extern "C" int trl_ppo_reduce_adam_net_f32(const float* partial, const double* scal_partial, int n_wg, int net,
                                           int D, int H, int A, float* grads, double* info, const trl_adam_t* adam,
                                           float* workspace, void* stream) {
  TRL_REQUIRE(net == 0 || net == 1, "net: 0 = policy, 1 = value function");
  return launch_reduce_adam(partial, scal_partial, n_wg, 0, D, H, A, grads, info, adam, workspace, nullptr, 0, stream, net);
}

// Env shards on several ranks: the same launch with the gradient SUM over ranks between the fold and the clip
// (torchrl/algo/on_policy/ppo.py:72-74, 117-119: clip_grad_norm_ sees the whole-minibatch gradient).  Needs a
// communicator whose peers are mapped (trl_comm_peer_open).
extern "C" int trl_ppo_reduce_adam_xrank_f32(const float* partial, const double* scal_partial, int n_wg, int n_wg_pf,
                                             int D, int H, int A, float* grads, double* info, const trl_adam_t* adam,
                                             float* workspace, trl_comm_t* comm, void* stream) {
  const XrArgs* xr = trl_comm_xr(comm);
  if (!xr) { trl_set_error("trl_ppo_reduce_adam_xrank_f32: communicator without mapped peers"); return TRL_EINVAL; }
  return launch_reduce_adam(partial, scal_partial, n_wg, n_wg_pf, D, H, A, grads, info, adam, workspace, xr,
                            trl_comm_wait_blocks(comm), stream);
}

// One network's half with the cross-rank SUM inside: the two update chains of a rank whose env shards sit on several
// ranks.  Each chain counts its own exchanges (ctl[4] / ctl[6] of the communicator) and owns its network's granules of the
// gradient region, so the two sequences of waiting launches never meet; a bounded wait footprint is shared between them.
extern "C" int trl_ppo_reduce_adam_xrank_net_f32(const float* partial, const double* scal_partial, int n_wg, int net,
                                                 int D, int H, int A, float* grads, double* info, const trl_adam_t* adam,
                                                 float* workspace, trl_comm_t* comm, void* stream) {
  TRL_REQUIRE(net == 0 || net == 1, "net: 0 = policy, 1 = value function");
  const XrArgs* xr = trl_comm_xr(comm);
  if (!xr) { trl_set_error("trl_ppo_reduce_adam_xrank_net_f32: communicator without mapped peers"); return TRL_EINVAL; }
  const int wb = trl_comm_wait_blocks(comm);
  return launch_reduce_adam(partial, scal_partial, n_wg, 0, D, H, A, grads, info, adam, workspace, xr,
                            wb > 0 ? (wb + 1) / 2 : 0, stream, net);
}

// The whole minibatch step as ONE launch (single process): trl_ppo_minibatch_grad_f32 + trl_ppo_reduce_adam_f32, bit for bit.
// The grid must be co-resident (n_wg <= CUs of the device, <= 256): its workgroups wait for each other inside the launch.
extern "C" int trl_ppo_step_workspace(int D, int H, int A) {
  const int ps = trl_ppo_partial_stride(D, H, A);
  if (ps < 0) return ps;
#ifdef STEP_CLK
  return step_ws_words(trl_ceil_div(ps, RED_CHUNK)) + 2 * 8 * STEP_FLAGS;      // + 8 phase stamps per workgroup
#endif
  return step_ws_words(trl_ceil_div(ps, RED_CHUNK));   // trl_ppo_reduce_adam_workspace's region first: the two routes share the Adam header
}
extern "C" int trl_ppo_step_max_workgroups(void) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  return cus < STEP_FLAGS ? cus : STEP_FLAGS;
}
extern "C" int trl_ppo_minibatch_step_f32(const trl_ppo_batch_t* p, float* grads, double* info, const trl_adam_t* adam,
                                          float* workspace, void* stream) {
  if (!p) { trl_set_error("ppo_step: null descriptor"); return TRL_EINVAL; }
  TRL_REQUIRE(grads && info && workspace, "null pointer");
  const int D = p->D, H = p->H, A = p->A;
  const int ps = trl_ppo_partial_stride(D, H, A);
  if (ps < 0) return ps;
  StepDev t{};
  int rc = fill_adam(adam, t.adam);
  if (rc) return rc;
  const int p_pf = H * D + H + H * H + H + A * H + A + A, p_vf = H * D + H + H * H + H + H + 1;
  TRL_REQUIRE(adam->n_groups == 2 && adam->group_sizes[0] == p_pf && adam->group_sizes[1] == p_vf,
              "optimiser groups must be [policy | value] of this shape");
  TRL_REQUIRE(adam->grads == grads, "adam->grads must be the fold output");
  TRL_REQUIRE(adam->params == p->pf_params && adam->params + p_pf == p->vf_params,
              "the networks' parameter blocks must be the optimiser's [policy | value] block");
  const int max_wg = trl_ppo_step_max_workgroups();
  if (p->n_wg > max_wg) {
    trl_set_error("ppo_step: %d workgroups cannot be co-resident on this device (%d): use the two-launch sequence", p->n_wg, max_wg);
    return TRL_EUNSUPPORTED;
  }
  t.grads = grads; t.info = info; t.ws = workspace; t.epoch = (unsigned)adam->step_count; t.device_state = adam->device_state;
  t.logstd = adam->params + (p_pf - A); t.n_act = A; t.p_pf = p_pf; t.p_vf = p_vf;
  return ppo_grad_launch(p, &t, stream);
}

#define TICK_PENDING 0x70000001                    /* clip_adam_launch: the step state still has to be advanced */
static int clip_adam_launch(const trl_adam_t* p, void* stream, bool tick_here);
extern "C" int trl_clip_adam_f32(const trl_adam_t* p, void* stream) { return clip_adam_launch(p, stream, true); }

// clip + Adam followed by the Polyak step of the target networks (rl_algo.py:169-176 / utils.py:16-20), two launches:
// when the device-resident step state needs the one-thread tick kernel (large parameter blocks), the Polyak kernel's
// first thread advances it instead -- it runs behind the Adam kernel in stream order, i.e. after every block has read it.
// `file` (optional): block 0 also archives the update's statistics block into row (updates finished before this one) mod
// slots of a ring -- the last launch of the update, every statistic has been written by then.
struct FileRing { const uint32_t* raw; uint32_t* ring; const double* count; int words, slots, ticked; };
__global__ __launch_bounds__(256) void polyak_tick_kernel(float* __restrict__ tgt, const float* __restrict__ src, int64_t n,
                                                          float tau, double* __restrict__ st, float beta1, float beta2,
                                                          FileRing file) {
  if (file.ring && blockIdx.x == 0) {
    const int64_t u = (int64_t)file.count[0] - file.ticked;            // (read by every thread BEFORE thread 0 advances it)
    __syncthreads();
    uint32_t* row = file.ring + (int64_t)(((u % file.slots) + file.slots) % file.slots) * file.words;
    for (int w = threadIdx.x; w < file.words; w += 256) row[w] = file.raw[w];
  }
  if (st && blockIdx.x == 0 && threadIdx.x == 0) { st[0] += 1.0; st[1] *= (double)beta1; st[2] *= (double)beta2; }
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
    tgt[e] = tgt[e] * (1.0f - tau) + src[e] * tau;
}
static int clip_adam_polyak(const trl_adam_t* p, float* target, const float* source, int64_t n, float tau,
                            const void* raw, int raw_bytes, void* ring, int slots, void* stream) {
  TRL_REQUIRE(n > 0 && target && source, "polyak: empty / null");
  int rc = clip_adam_launch(p, stream, false);
  if (rc != TRL_OK && rc != TICK_PENDING) return rc;
  int grid = trl_ceil_div(n, 256);
  if (grid > 1024) grid = 1024;
  FileRing file = {(const uint32_t*)raw, (uint32_t*)ring, p->step_state, raw_bytes / 4, slots, rc == TICK_PENDING ? 0 : 1};
  hipLaunchKernelGGL(polyak_tick_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, target, source, n, tau,
                     rc == TICK_PENDING ? p->step_state : (double*)nullptr, p->beta1, p->beta2, file);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}
extern "C" int trl_clip_adam_polyak_f32(const trl_adam_t* p, float* target, const float* source, int64_t n, float tau,
                                        const void* raw, int raw_bytes, void* ring, int slots, void* stream) {
  if (!ring) return clip_adam_polyak(p, target, source, n, tau, nullptr, 0, nullptr, 0, stream);
  TRL_REQUIRE(raw && raw_bytes > 0 && raw_bytes % 4 == 0 && slots > 0, "clip_adam_polyak: bad ring");
  TRL_REQUIRE(p && p->step_state, "clip_adam_polyak: the ring row is picked by the device-resident step state");
  return clip_adam_polyak(p, target, source, n, tau, raw, raw_bytes, ring, slots, stream);
}

// ---------------------------------------------------------------- fold + clip + Adam + Polyak, ONE launch
// The tail of an off-policy update (twin_sac_q.py:162-220): fold of the split weight-gradient partials of every layer
// (trl_fold_partials_multi_f32's arithmetic and summation order), clip_grad_norm_ per network, the Adam steps, the Polyak
// step of the target networks (algo/utils.py:16-20), the filing of the update's statistics block -- three launches before
// (7 + 5 + 5 us for 217 k parameters at cfg 3, each latency-bound).  A grid of <= 256 always co-resident workgroups: a
// thread folds its parameters and leaves them in `grads`, the workgroup publishes its share of each group's sum of squares as
// {epoch, value} granules, every workgroup polls all of them (the rendezvous of ppo_reduce_adam_kernel: no ticket, no
// reset, device-scope relaxed accesses only), derives the clip coefficients in the same fixed order and steps its own
// parameters and their targets.  The step state is read by every workgroup BEFORE it publishes; workgroup 0 advances it
// after it has seen every granule.  A poll that does not complete (~1 s) trips workspace word 0 and leaves the parameters
// untouched.
#define FAP_THREADS 1024
#define FAP_MAX_WG 256
#define FAP_MAX_ENTRIES 32
struct FapDev {
  int count;
  int off[FAP_MAX_ENTRIES + 1];                    // entry k owns grads[off[k] .. off[k + 1])
  int splits[FAP_MAX_ENTRIES];
  const float* part[FAP_MAX_ENTRIES];              // [splits][n_k]
  float* target; int target_off, target_n; float tau;
  unsigned long long* slots;                       // [4 groups][FAP_MAX_WG] {epoch, ss bits}
  unsigned* err;
  float* grads_out;
};
__global__ __launch_bounds__(FAP_THREADS) void fold_adam_polyak_kernel(AdamDev a, FapDev f, FileRing file) {
  __shared__ float s_ss[4][FAP_THREADS / 64];
  __shared__ float s_coef[4];
  __shared__ float s_bc[2];
  __shared__ int s_fail;
  // (the entry table goes to LDS: a per-lane index into the argument struct would be a dependent load per element)
  __shared__ int s_off[FAP_MAX_ENTRIES + 1], s_splits[FAP_MAX_ENTRIES];
  __shared__ const float* s_part[FAP_MAX_ENTRIES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nb = gridDim.x;
  if (tid <= f.count) s_off[tid] = f.off[tid];
  if (tid < f.count) { s_splits[tid] = f.splits[tid]; s_part[tid] = f.part[tid]; }
  const int total = a.off[a.n_groups];
  const int per = (total + nb - 1) / nb, lo = blockIdx.x * per, hi = min(total, lo + per);
  // the step state and the ring row are read before anything is published (workgroup 0 advances the state at the end)
  const double steps = a.step_state[0];
  const unsigned epoch = (unsigned)(long long)steps + 1u;
  if (tid == 0) {
    s_bc[0] = (float)(1.0 - a.step_state[1] * (double)a.beta1);
    s_bc[1] = (float)sqrt(1.0 - a.step_state[2] * (double)a.beta2);
    s_fail = 0;
  }
  __syncthreads();
  // ---- fold (trl_fold_partials_multi_f32's order: four interleaved chains over the splits, then (0 + 1) + (2 + 3)) ----
  float ss[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int e = lo + tid; e < hi; e += FAP_THREADS) {
    int k = 0;
    while (k + 1 < f.count && e >= s_off[k + 1]) ++k;
    const int n = s_off[k + 1] - s_off[k], splits = s_splits[k];
    const float* p = s_part[k] + (e - s_off[k]);
    float c[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int s0 = 0;
    for (; s0 + 16 <= splits; s0 += 16) {          // 16 loads in flight: chain j takes splits j, j + 4, j + 8, ...
      float x[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) x[q] = p[(size_t)(s0 + q) * n];
#pragma unroll
      for (int q = 0; q < 16; ++q) c[q & 3] += x[q];
    }
    for (; s0 < splits; ++s0) c[s0 & 3] += p[(size_t)s0 * n];
    const float gsum = (c[0] + c[1]) + (c[2] + c[3]);
    f.grads_out[e] = gsum;
    int g = 0;
    while (e >= a.off[g + 1]) ++g;
    const float gs = gsum * a.grad_scale;
#pragma unroll
    for (int q = 0; q < 4; ++q) ss[q] = fmaf(q == g ? gs : 0.0f, gs, ss[q]);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float w = wave_sum(ss[q]);
    if (lane == 0) s_ss[q][wave] = w;
  }
  __syncthreads();
  if (tid < a.n_groups) {
    float t = 0.0f;
    for (int w = 0; w < FAP_THREADS / 64; ++w) t += s_ss[tid][w];
    __hip_atomic_store(f.slots + tid * FAP_MAX_WG + blockIdx.x, ((unsigned long long)epoch << 32) | __float_as_uint(t),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- rendezvous: every workgroup's granules of this launch; group norms in workgroup order ----
  // wave w polls granules 64 r .. 64 r + 63 of group g (w = 4 g + r): all rounds of all groups fly together
  {
    const int g = wave >> 2, r = wave & 3, b = 64 * r + lane;
    float v = 0.0f;
    if (g < a.n_groups && b < nb) {
      const unsigned long long* slot = f.slots + g * FAP_MAX_WG + b;
      unsigned long long x = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long t0 = wall_clock64();
      while ((unsigned)(x >> 32) != epoch) {
        __builtin_amdgcn_s_sleep(1);
        x = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wall_clock64() - t0 > 100000000ull) { s_fail = 1; break; }
      }
      v = __uint_as_float((unsigned)x);
    }
    v = wave_sum(v);                                   // a fixed tree over the round's 64 values
    if (lane == 0) s_ss[g & 3][r] = v;                 // (s_ss is free again: its sums were published above)
  }
  __syncthreads();
  if (tid < a.n_groups) {
    const float t = ((s_ss[tid][0] + s_ss[tid][1]) + s_ss[tid][2]) + s_ss[tid][3];      // rounds in order
    const float norm = sqrtf(t);
    s_coef[tid] = (a.max_norm > 0.0f) ? fminf(a.max_norm / (norm + 1e-6f), 1.0f) : 1.0f;
    if (blockIdx.x == 0 && a.norms_out) a.norms_out[tid] = norm;
  }
  __syncthreads();
  if (s_fail) { if (tid == 0) __hip_atomic_store(f.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
  // ---- Adam on the own parameters, Polyak on their targets ----
  const float bc1 = s_bc[0], bc2_sqrt = s_bc[1];
  for (int e = lo + tid; e < hi; e += FAP_THREADS) {
    int g = 0;
    while (e >= a.off[g + 1]) ++g;
    const float gr = f.grads_out[e] * a.grad_scale * s_coef[g];
    const float m = a.beta1 * a.m[e] + (1.0f - a.beta1) * gr;
    const float v = a.beta2 * a.v[e] + (1.0f - a.beta2) * gr * gr;
    a.m[e] = m; a.v[e] = v;
    const float denom = sqrtf(v) / bc2_sqrt + a.eps;
    const float lr = a.device_lr ? a.device_lr[g] : a.lr[g];
    const float pn = a.params[e] - (lr / bc1) * (m / denom);
    a.params[e] = pn;
    const int te = e - f.target_off;
    if (te >= 0 && te < f.target_n) f.target[te] = f.target[te] * (1.0f - f.tau) + pn * f.tau;
  }
  if (blockIdx.x == 0) {
    if (file.ring) {
      const int64_t u = (int64_t)steps;
      uint32_t* row = file.ring + (int64_t)(((u % file.slots) + file.slots) % file.slots) * file.words;
      for (int w = tid; w < file.words; w += FAP_THREADS) row[w] = file.raw[w];
    }
    if (tid == 0) { a.step_state[0] = steps + 1.0; a.step_state[1] *= (double)a.beta1; a.step_state[2] *= (double)a.beta2; }
  }
}
extern "C" int trl_fold_clip_adam_polyak_workspace(void) { return (int)(16 + 4 * FAP_MAX_WG * sizeof(unsigned long long)); }   // bytes, zeroed once
extern "C" int trl_fold_clip_adam_polyak_f32(int count, const float* const* part, const int* n, const int* splits,
                                             const trl_adam_t* adam, float* target, int64_t target_off, int64_t target_n,
                                             float tau, const void* raw, int raw_bytes, void* ring, int slots,
                                             void* workspace, void* stream) {
  TRL_REQUIRE(count >= 1 && count <= FAP_MAX_ENTRIES && part && n && splits, "fold_clip_adam_polyak: 1..32 fold entries");
  TRL_REQUIRE(adam && adam->step_state && workspace, "fold_clip_adam_polyak: needs the device-resident step state and a workspace");
  TRL_REQUIRE((!target_n) || (target && target_off >= 0), "fold_clip_adam_polyak: bad target range");
  TRL_REQUIRE(!ring || (raw && raw_bytes > 0 && raw_bytes % 4 == 0 && slots > 0), "fold_clip_adam_polyak: bad ring");
  AdamDev d;
  int rc = fill_adam(adam, d);
  if (rc) return rc;
  FapDev f{};
  f.count = count;
  for (int k = 0; k < count; ++k) {
    TRL_REQUIRE(part[k] && n[k] > 0 && splits[k] >= 1, "fold_clip_adam_polyak: bad entry");
    f.part[k] = part[k]; f.splits[k] = splits[k]; f.off[k + 1] = f.off[k] + n[k];
  }
  const int total = d.off[d.n_groups];
  TRL_REQUIRE(f.off[count] == total, "fold_clip_adam_polyak: the fold entries must cover the parameter block exactly, in order");
  TRL_REQUIRE(target_off + target_n <= total, "fold_clip_adam_polyak: target range outside the parameter block");
  f.target = target; f.target_off = (int)target_off; f.target_n = (int)target_n; f.tau = tau;
  f.err = (unsigned*)workspace;
  f.slots = (unsigned long long*)((char*)workspace + 16);
  f.grads_out = const_cast<float*>(adam->grads);
  FileRing file = {(const uint32_t*)raw, (uint32_t*)ring, adam->step_state, raw_bytes / 4, slots, 0};
  const int grid = std::max(1, std::min(FAP_MAX_WG, trl_ceil_div(total, FAP_THREADS)));
  hipLaunchKernelGGL(fold_adam_polyak_kernel, dim3(grid), dim3(FAP_THREADS), 0, (hipStream_t)stream, d, f, file);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

static int clip_adam_launch(const trl_adam_t* p, void* stream, bool tick_here) {
  AdamDev d;
  int rc = fill_adam(p, d);
  if (rc) return rc;
  TRL_REQUIRE(!p->device_state, "device_state is a feature of trl_ppo_reduce_adam_f32");
  const int total = d.off[p->n_groups];
  if (total == 0) return TRL_OK;
  // Up to 128 blocks count themselves through one atomic (a few hundred ns, hidden under the norm pass); larger
  // parameter blocks would serialise on it (860 blocks: 25 us measured), so they keep the one-thread tick launch.
  const int blocks = trl_ceil_div(total, 256);
  // (counting RETIRED blocks with a relaxed add at the end of each block instead was measured too: SAC cfg 3, 852 blocks,
  // 0.392 ms per update against 0.381 ms with the tick launch)
  d.self_tick = (d.step_state && blocks <= 128) ? 1 : 0;
  hipLaunchKernelGGL(clip_adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d);
  TRL_LAUNCH_CHECK();
  if (d.step_state && !d.self_tick) {
    if (!tick_here) return TICK_PENDING;             // the caller's next kernel advances the state
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, d.step_state, d.beta1, d.beta2);
    TRL_LAUNCH_CHECK();
  }
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

extern "C" int trl_mlp2_forward_supported(int D, int H, int O) {
  return (D == 17 && H == 64 && (O == 6 || O == 1)) ? 1 : 0;     // keep in step with the dispatch below
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
