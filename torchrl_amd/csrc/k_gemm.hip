// K10 (generic) -- fp32 MFMA GEMM family for dense layers of any shape.
//
// The fused PPO kernels cover the benchmark network; everything else that is a dense layer in the
// reference (torchrl/networks/base.py:30-44, nets.py:34-52: nn.Linear + activation; 256-wide SAC
// nets, Q nets on [obs, act], FC heads of the conv nets) goes through this family:
//     forward   Y[M,N]  = act( X[M,K] . W[N,K]^T + b[N] )          (trans_b)
//     input-grad dX[M,K] = dZ[M,N] . W[N,K]                          (plain)
//     weight-grad dW[N,K] = dZ[M,N]^T . X[M,K]   (split over M, deterministic two-pass fold)
//     dZ = dY * act'(Y)  and  db = column sums of dZ are fused into the operand loads / a side output.
// The conv layers of CNNBase (networks/base.py:59-107) run through the same kernel as IMPLICIT GEMMs (template
// parameter CONV, entry points trl_conv_*): the im2col matrix is only ever an address computation.
// Exact fp32 on v_mfma_f32_32x32x2_f32: a workgroup of 4 waves owns a 64x64, 128x32 or 32x128 C tile (one 32x32
// quadrant per wave; template parameter WM).  These layers are a few thousand rows by <= 256..512 columns: the reduction is short, so what paces
// the kernel is how often it waits for memory, not the MFMAs.  The reduction is therefore walked in PANELS of
// KC = 128: a panel of each operand (64 x 128 floats) is fetched with 16-byte loads into registers while the
// previous panel's 64 MFMAs per wave run from LDS (one memory round trip per 128 k, all of a panel's loads in
// flight together; the first version walked K in steps of 32 and paid a round trip per step, 4 us per step at
// K = 512 where every workgroup's 128-byte row segments also sat 2 KB apart).  LDS tiles keep the layout the
// operand has in memory -- [row][k] (row stride KC + 4, read with one ds_read_b128 per 4 MFMAs) when k is the
// contiguous dimension, [k][col] (row stride rows + 8, ds_read_b32) when it is not -- so the global->LDS copy
// never transposes.  MFMA step (q, r) of lane half `hi` takes k = 8q + 4hi + r.  Arbitrary M, N, K: slots that
// are misaligned or cross an edge fall back to predicated scalar loads (zero fill).
#include <algorithm>
#include <stdlib.h>
#include "trl_common.h"
#include "trl_mlp.h"
#include "trl_conv.h"

#define KC 128                      // granule of split-reduction lengths (a multiple of every panel size below)

// act'(y) expressed through the activation OUTPUT y (tanh: 1 - y^2, relu: y > 0, none: 1)
__device__ __forceinline__ float dact_from_out(int act, float y) {
  if (act == TRL_ACT_TANH) return 1.0f - y * y;
  if (act == TRL_ACT_RELU) return y > 0.0f ? 1.0f : 0.0f;
  return 1.0f;
}

// Up to 8 independent problems of identical shape in one launch (blockIdx.y): the twin critics and their targets,
// one network on several inputs.  A 4096 x 256 x 256 layer is 256 workgroups and ~4 us of MFMA behind ~4 us of
// launch + prologue + epilogue; grouped, the problems' workgroups overlap each other's fixed costs.
#define GEMM_MAX_GROUPS 12
struct GemmGroup { const float* A; const float* B; float* C; const float* bias; const float* a_gate; float* colsum; };
// Problems of DIFFERENT shapes in one launch (the weight gradients of every layer of a backward pass: nothing waits
// for them until the optimiser step, so they need not be six dependent launches of 32 - 256 workgroups each)
struct GemmShape { int M, N, K, lda, ldb, ldc, split_len, tiles_n, tiles, splits; };

struct GemmDev {
  const float* A; const float* B; float* C;
  const float* bias;          // epilogue: + bias[n]          (nullable)
  const float* a_gate;        // operand A is A * act'(a_gate) elementwise, same shape/ld as A (nullable)
  int M, N, K, lda, ldb, ldc;
  int act;                    // epilogue activation (TRL_ACT_*), applied after bias
  int gate_act;               // activation whose derivative gates A
  int split_len;              // TA only: rows of the reduction handled by one blockIdx.z
  float* colsum;              // TA only: (splits, M) partial column sums of the gated A (nullable)
  int tiles_n, tiles;         // C tiles along N, and in total
  ConvSrc cv;                 // CONV != 0: the implicit operand (A when CONV == 1, B when CONV == 2)
  int chw_p;                  // CONV == 3, != 0: C is stored (b, n, p) with p = m % chw_p (= Ho * Wo), nn.Flatten's order
  int groups;                 // > 1: the operand pointers of problem blockIdx.y come from grp[]
  GemmGroup grp[GEMM_MAX_GROUPS];
  int prefer128;              // != 0: the caller sized its reduction split for 128 x 128 tiles (split-K forward of few-row layers)
  int hetero;                 // != 0: problem blockIdx.y also has its own shape (the grid is sized for the largest)
  int b_perm;                 // CONV == 3: B is already stored in the (i, j, c) reduction order (dense rows of K floats)
  GemmShape shp[GEMM_MAX_GROUPS];
};

// development aid (tools/bench_gemm.py --clk): shader-clock and 100 MHz real-time stamps of a few workgroups
#ifdef TRL_EXP_CLK
__device__ long long g_gemm_clk[8 * 16];
#define GCLK(ph) if (tid == 0 && (blockIdx.x & 31) == 0 && blockIdx.x < 256 && blockIdx.z == 0) { \
    g_gemm_clk[(blockIdx.x >> 5) * 16 + 2 * (ph)] = clock64(); g_gemm_clk[(blockIdx.x >> 5) * 16 + 2 * (ph) + 1] = wall_clock64(); }
extern "C" int trl_dbg_gemm_clk(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_clk), sizeof(long long) * 8 * 16);
}
#else
#define GCLK(ph)
#endif

// One operand panel: ROWS "rows" (the operand's C-side index; 32, 64 or 128) x KP reduction indices (the panel size of
// the instantiation: 64, or 32 for the 128 x 128 tile), held as ROWS * KP / 1024 16-byte slots per thread.  CONTIG_K:
// element (r, k) lives at base[(row0 + r) * ld + k0 + k] and the LDS tile is [ROWS][KP + 4] (16-byte aligned rows, banks
// rotate by 4 per row); otherwise it lives at base[(k0 + k) * ld + row0 + r] and the tile is [KP][ROWS + 8].  Slot t of
// thread tid covers 4 consecutive elements of the contiguous dim.
template <int KP> struct PanelGeom {
  static constexpr int LD_K = KP + 4;               // row stride of a [row][k] tile
  static constexpr int KT = KP / 4;                 // threads per row of a [row][k] tile
  static constexpr int RP = 256 / KT;               // rows covered by one slot round of the 256 threads
};
template <bool CONTIG_K, int ROWS, int KP> __host__ __device__ constexpr int panel_slots() { return ROWS * KP / 1024; }
template <bool CONTIG_K, int ROWS, int KP>
__device__ __forceinline__ void panel_slot(int tid, int t, int& r, int& k) {
  using G = PanelGeom<KP>;
  constexpr int TPR = ROWS / 4;                    // threads per reduction row of a [k][row] tile
  if (CONTIG_K) { r = G::RP * t + tid / G::KT; k = 4 * (tid % G::KT); }
  else          { k = (256 / TPR) * t + tid / TPR; r = 4 * (tid % TPR); }
}

// whole panel in range and 16-byte loads legal: unconditional loads a fixed stride apart
template <bool CONTIG_K, int ROWS, int KP>
__device__ __forceinline__ void panel_fetch_fast(const float* __restrict__ base, int ld, int row0, int k0, int tid,
                                                 f32x4 (&reg)[ROWS * KP / 1024]) {
  int r, k;
  panel_slot<CONTIG_K, ROWS, KP>(tid, 0, r, k);
  const float* p = CONTIG_K ? base + (size_t)(row0 + r) * ld + k0 + k : base + (size_t)(k0 + k) * ld + row0 + r;
  const size_t step = (size_t)(CONTIG_K ? PanelGeom<KP>::RP : 256 / (ROWS / 4)) * ld;
#pragma unroll
  for (int t = 0; t < ROWS * KP / 1024; ++t) reg[t] = *reinterpret_cast<const f32x4*>(p + t * step);
}

// edge / misaligned panel: element-wise predicated loads, zero fill
template <bool CONTIG_K, int ROWS, int KP>
__device__ __forceinline__ void panel_fetch_edge(const float* __restrict__ base, int ld, int row0, int rows, int k0, int k_hi,
                                                 int tid, f32x4 (&reg)[ROWS * KP / 1024]) {
#pragma unroll
  for (int t = 0; t < ROWS * KP / 1024; ++t) {
    int r, k;
    panel_slot<CONTIG_K, ROWS, KP>(tid, t, r, k);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gr = row0 + r + (CONTIG_K ? 0 : j), gk = k0 + k + (CONTIG_K ? j : 0);
      const bool ok = gr < rows && gk < k_hi;
      const size_t idx = CONTIG_K ? (size_t)gr * ld + gk : (size_t)gk * ld + gr;
      reg[t][j] = ok ? base[ok ? idx : 0] : 0.0f;
    }
  }
}

template <bool CONTIG_K, int ROWS, int KP>
__device__ __forceinline__ void panel_stash(float* tile, int tid, const f32x4 (&reg)[ROWS * KP / 1024]) {
#pragma unroll
  for (int t = 0; t < ROWS * KP / 1024; ++t) {
    int r, k;
    panel_slot<CONTIG_K, ROWS, KP>(tid, t, r, k);
    *reinterpret_cast<f32x4*>(tile + (CONTIG_K ? r * PanelGeom<KP>::LD_K + k : k * (ROWS + 8) + r)) = reg[t];
  }
}

// the 4 operand values of MFMA steps (q, 0..3) for C-side index `row` of this lane half
template <bool CONTIG_K, int ROWS, int KP>
__device__ __forceinline__ f32x4 panel_operand(const float* tile, int row, int q, int hi) {
  if (CONTIG_K) return *reinterpret_cast<const f32x4*>(tile + row * PanelGeom<KP>::LD_K + 8 * q + 4 * hi);
  constexpr int LD = ROWS + 8;                     // rows k and k + 4 (the two lane halves) are 32 banks apart
  const float* p = tile + (8 * q + 4 * hi) * LD + row;
  f32x4 v = {p[0], p[LD], p[2 * LD], p[3 * LD]};
  return v;
}

template <bool CONTIG_K, int ROWS, int KP>
__host__ __device__ constexpr int tile_floats() { return CONTIG_K ? ROWS * (KP + 4) : KP * (ROWS + 8); }

// panel size of an instantiation: 64 reduction indices (32 MFMAs per wave and barrier), 32 for the 2 x 2-block wave tile
// (64 MFMAs per wave and barrier) -- two panel buffers of either fit twice into a CU's 160 KB (two workgroups per CU)
template <int TMB> __host__ __device__ constexpr int gemm_panel() { return TMB == 2 ? 32 : 64; }

// TA: A is stored [Kred][M] (we need A^T); TB: B is stored [N][K] (we need B^T).
// GATE: activation whose derivative (through a_gate) multiplies operand A (TRL_ACT_NONE: no gate).
// CONV: 0 both operands dense; 1 operand A (M x K, forward) is the implicit cols matrix of g.cv over uint8 NCHW
// frames; 2 operand B (Kred x N, weight gradient) is; 3 / 4 the same over fp32 NHWC activations with the
// reduction (3) or column (4) index in (i, j, c) order -- then B of 3 is the permuted view of the conv weight.
// WM: waves along M.  TMB: 32 x 32 accumulator blocks per wave along each of M and N.  With TMB = 1 the 4 waves (one
// quadrant each) form a 64 x 64 C tile (WM = 2), a 128 x 32 one (WM = 4, layers with <= 32 outputs: a 64-wide tile would
// compute 50-75 % padding) or a 32 x 128 one (WM = 1); TMB = 2 (WM = 2 only) is the 128 x 128 tile of large products:
// every operand value read from LDS feeds two MFMAs instead of one.
//
// Main loop (round 4): TWO panel buffers in LDS and ONE barrier per panel.  While the MFMAs of panel p run from buffer
// p & 1, the registers that hold panel p + 1 (fetched one panel earlier) are written to the other buffer and the loads
// of panel p + 2 are issued -- the register -> LDS hand-off and the global round trip both sit in the shadow of the
// matrix pipe, where round 3's single buffer had `stash; barrier; fetch; MFMA; barrier` in series (0.64 of the measured
// MFMA peak at 4096^3).  The k order of every output element is unchanged (panels ascending, k = 8 q + 4 hi + r inside),
// so results are bit-identical to the one-buffer loop.
// KPX (0: the tile's default): panel size override.  KPX = 32 on the 64 x 64 tile halves the two panel buffers to 36 KB,
// so FOUR workgroups share a CU instead of two -- for reductions of a few panels per workgroup (K = 256 layers, 256-row
// slices of a split weight gradient) the prologue / epilogue of one workgroup then hides under the MFMAs of three others.
template <bool TA, bool TB, int GATE, int CONV, int WM, int TMB = 1, int KPX = 0>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmDev g) {
  static_assert(CONV == 0 || ((CONV == 1 || CONV == 3) && !TA && TB) || ((CONV == 2 || CONV == 4) && TA && !TB),
                "implicit operand orientation");
  static_assert(TMB == 1 || (TMB == 2 && WM == 2 && CONV == 0), "the 2 x 2-block wave tile is dense and square");
  constexpr bool CA = CONV == 1 || CONV == 3, CB = CONV == 2 || CONV == 4, U8 = CONV == 1 || CONV == 2;
  static_assert(KPX == 0 || (KPX == 32 && (CONV == 0 || CONV == 3)), "panel override: 32; dense operands or the fp32 channels-last forward");
  constexpr int KP = KPX ? KPX : gemm_panel<TMB>(), NQ = KP / 8;
  using PG = PanelGeom<KP>;
  constexpr int WN = 4 / WM, GM = 32 * WM * TMB, GN = 32 * WN * TMB;
  constexpr int SA = panel_slots<!TA, GM, KP>(), SB = panel_slots<TB, GN, KP>();
  constexpr int AF = tile_floats<!TA, GM, KP>(), BF = tile_floats<TB, GN, KP>();
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Workgroups are dealt round-robin to the 8 XCDs: renumber so that each XCD owns a contiguous run of tiles
  // (the tiles_n tiles that share an A panel then share an L2).
  // the problem's shape: the launch's, or (hetero) this problem's own -- surplus workgroups of the common grid leave.
  // (Locals, not writes into `g`: a modified kernel-argument struct is demoted to scratch memory.)
  int g_M = g.M, g_N = g.N, g_K = g.K, g_lda = g.lda, g_ldb = g.ldb, g_ldc = g.ldc, g_split_len = g.split_len,
      g_tiles_n = g.tiles_n, g_tiles = g.tiles;
  if (g.hetero) {
    const GemmShape& h = g.shp[blockIdx.y];
    if ((int)blockIdx.x >= h.tiles || (int)blockIdx.z >= h.splits) return;
    g_M = h.M; g_N = h.N; g_K = h.K; g_lda = h.lda; g_ldb = h.ldb; g_ldc = h.ldc;
    g_split_len = h.split_len; g_tiles_n = h.tiles_n; g_tiles = h.tiles;
  }
  int tile = blockIdx.x;
  if ((g_tiles & 7) == 0) tile = (tile & 7) * (g_tiles >> 3) + (tile >> 3);
  const int tm = tile / g_tiles_n, tn = tile - tm * g_tiles_n;
  const int m0 = tm * GM, n0 = tn * GN;
  int k_lo = 0, k_hi = g_K;
  const float* pA = g.A; const float* pB = g.B; float* C = g.C;
  const float* pBias = g.bias; const float* pGate = g.a_gate; float* pColsum = g.colsum;
  const float* cvx = g.cv.x;                       // CONV 3: the activations the implicit A operand is read from
  if (g.groups > 1) {                              // grouped launch: problem blockIdx.y
    const GemmGroup& q = g.grp[blockIdx.y];
    pA = q.A; pB = q.B; C = q.C; pBias = q.bias; pGate = q.a_gate; pColsum = q.colsum;
    if (CONV == 3) cvx = q.A;                      // (a grouped implicit forward carries its input in the free A slot)
  }
  if (gridDim.z > 1) {                             // split reduction: blockIdx.z owns split_len indices, writes its own
    k_lo = blockIdx.z * g_split_len;               // partial C (folded in fixed order by fold_partials_kernel)
    k_hi = min(g_K, k_lo + g_split_len);
    C += (size_t)blockIdx.z * g_M * g_ldc;
  }
  const int wm = wave / WN, wn = wave % WN;
  const int i = lane & 31, hi = lane >> 5;
  const bool a_whole = (g_lda & 3) == 0 && (reinterpret_cast<uintptr_t>(pA) & 15) == 0 &&
                       (GATE == TRL_ACT_NONE || (reinterpret_cast<uintptr_t>(pGate) & 15) == 0) && m0 + GM <= g_M;
  const bool b_whole = (g_ldb & 3) == 0 && (reinterpret_cast<uintptr_t>(pB) & 15) == 0 && n0 + GN <= g_N;
  const bool want_colsum = TA && pColsum && tn == 0;

  f32x16 acc[TMB][TMB];
#pragma unroll
  for (int a = 0; a < TMB; ++a)
#pragma unroll
    for (int b = 0; b < TMB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
  f32x4 csum = {0.0f, 0.0f, 0.0f, 0.0f};          // TA: partial column sums of columns 4 * (tid % (GM / 4)) .. + 3
  f32x4 ra[SA], rg[SA], rb[SB];
  // Implicit operand.  fp32 NHWC: one 16-byte load per slot, same slot map as a dense panel (consecutive lanes read
  // consecutive pieces of a window row).  uint8 NCHW: one dword per slot with LANES ALONG THE OUTPUT POSITIONS --
  // neighbouring windows are sw bytes apart, so a wave's load is (nearly) contiguous, where the dense slot map
  // would gather 64 scattered 4-byte pieces; thread = one operand row, its slots walk the other dimension.
  constexpr int SC = CA ? SA : SB;
  uint32_t cu[SC], cmask = 0u, cbase[SC], ctap = 0u;
  bool ctap_ok = false;
  uint32_t* tap_tab = reinterpret_cast<uint32_t*>(lds + 2 * (AF + BF));          // CONV == 1: tap offset of every k / 4
  constexpr int GA = 256 / GM;                     // CONV == 1: thread = row tid % GM, slots k4 = tid / GM + GA * t
  constexpr int GB = 256 / KP;                     // CONV == 2: thread = reduction row tid % KP, slots c4 = tid / KP + GB * t
  if (CONV == 1) {
    const int m = m0 + tid % GM;
    cbase[0] = m < g_M ? conv_row_offset(g.cv, (uint32_t)m) : 0xffffffffu;
    for (int e = tid; 4 * e < g_K; e += 256) tap_tab[e] = conv_tap_offset(g.cv, (uint32_t)(4 * e));
    __syncthreads();
  }
  if (CONV == 2) {
#pragma unroll
    for (int t = 0; t < SC; ++t) {
      const int kc = n0 + 4 * (tid / KP + GB * t);
      cbase[t] = kc < g_N ? conv_tap_offset(g.cv, (uint32_t)kc) : 0xffffffffu;   // here: the (fixed) tap offsets
    }
  }
  if (CONV == 3) {                                 // A rows are fixed for the whole kernel: decode them once
#pragma unroll
    for (int t = 0; t < SC; ++t) {
      const int m = m0 + PG::RP * t + tid / PG::KT;
      cbase[t] = m < g_M ? nhwc_row_offset(g.cv, (uint32_t)m) : 0xffffffffu;
    }
  }
  if (CONV == 4) {                                 // B columns (the taps) are fixed for the whole kernel
    const int kc = n0 + 4 * (tid % (GN / 4));
    ctap_ok = kc < g_N;
    ctap = ctap_ok ? nhwc_tap_offset(g.cv, (uint32_t)kc) : 0u;
  }
  const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};

  auto fetch = [&](int k0) {
    const bool k_whole = k0 + KP <= k_hi;          // uniform: one branch per operand per panel
    if (CONV == 1) {
      cmask = 0u;
#pragma unroll
      for (int t = 0; t < SC; ++t) {
        const int k4 = tid / GM + GA * t;
        const bool ok = k0 + 4 * k4 < k_hi && cbase[0] != 0xffffffffu;
        cu[t] = ok ? *reinterpret_cast<const uint32_t*>(g.cv.frames + (ok ? cbase[0] + tap_tab[(k0 >> 2) + k4] : 0u)) : 0u;
        cmask |= ok ? (1u << t) : 0u;
      }
    } else if (CONV == 3) {
      const int kk = k0 + 4 * (tid % PG::KT);
      const bool kin = kk < k_hi;
      const uint32_t tap = kin ? nhwc_tap_offset(g.cv, (uint32_t)kk) : 0u;
#pragma unroll
      for (int t = 0; t < SC; ++t) {
        const bool ok = kin && cbase[t] != 0xffffffffu;
        ra[t] = ok ? *reinterpret_cast<const f32x4*>(cvx + (ok ? cbase[t] + tap : 0u)) : zero4;
      }
    } else if (a_whole && k_whole) {
      panel_fetch_fast<!TA, GM, KP>(pA, g_lda, m0, k0, tid, ra);
      if (GATE != TRL_ACT_NONE && pGate) panel_fetch_fast<!TA, GM, KP>(pGate, g_lda, m0, k0, tid, rg);
    } else {
      panel_fetch_edge<!TA, GM, KP>(pA, g_lda, m0, g_M, k0, k_hi, tid, ra);
      if (GATE != TRL_ACT_NONE && pGate) panel_fetch_edge<!TA, GM, KP>(pGate, g_lda, m0, g_M, k0, k_hi, tid, rg);
    }
    if (CONV == 2) {
      const int m = k0 + tid % KP;
      const bool row_ok = m < k_hi;
      const uint32_t row = row_ok ? conv_row_offset(g.cv, (uint32_t)m) : 0u;
      cmask = 0u;
#pragma unroll
      for (int t = 0; t < SC; ++t) {
        const bool ok = row_ok && cbase[t] != 0xffffffffu;
        cu[t] = ok ? *reinterpret_cast<const uint32_t*>(g.cv.frames + (ok ? row + cbase[t] : 0u)) : 0u;
        cmask |= ok ? (1u << t) : 0u;
      }
    } else if (CONV == 4) {
#pragma unroll
      for (int t = 0; t < SC; ++t) {
        const int m = k0 + (256 / (GN / 4)) * t + tid / (GN / 4);
        const bool ok = ctap_ok && m < k_hi;
        rb[t] = ok ? *reinterpret_cast<const f32x4*>(g.cv.x + (ok ? nhwc_row_offset(g.cv, (uint32_t)m) + ctap : 0u)) : zero4;
      }
    } else if (CONV == 3 && !g.b_perm) {
      // B = conv weight (Cout, C, kh, kw) read in the reduction order k' = (i, j, c): 4 consecutive k' are 4
      // consecutive channels of one tap, kh * kw floats apart
      const int kk = k0 + 4 * (tid % PG::KT);
      const uint32_t khw = (uint32_t)(g.cv.kh * g.cv.kw);
      const uint32_t ij = (uint32_t)kk / (uint32_t)g.cv.C, c = (uint32_t)kk - ij * g.cv.C;
      const uint32_t w0 = c * khw + ij;
#pragma unroll
      for (int t = 0; t < SB; ++t) {
        const int n = n0 + PG::RP * t + tid / PG::KT;
        const bool ok = kk < k_hi && n < g_N;
        const float* wp = pB + (ok ? (size_t)n * g_ldb + w0 : 0);
        f32x4 v = {ok ? wp[0] : 0.0f, ok ? wp[khw] : 0.0f, ok ? wp[2 * khw] : 0.0f, ok ? wp[3 * khw] : 0.0f};
        rb[t] = v;
      }
    } else if (b_whole && k_whole) panel_fetch_fast<TB, GN, KP>(pB, g_ldb, n0, k0, tid, rb);
    else                           panel_fetch_edge<TB, GN, KP>(pB, g_ldb, n0, g_N, k0, k_hi, tid, rb);
  };
  auto stash = [&](int buf) {
    float* As = lds + buf * (AF + BF);
    float* Bs = As + AF;
    if (CONV == 1) {                               // uint8 slots go straight to their (row, k4) place
#pragma unroll
      for (int t = 0; t < SC; ++t) {
        const f32x4 v = (cmask >> t) & 1u ? conv_unpack(cu[t], g.cv.scale, g.cv.shift) : zero4;
        *reinterpret_cast<f32x4*>(As + (tid % GM) * PG::LD_K + 4 * (tid / GM + GA * t)) = v;
      }
    }
    if (CONV == 2) {
#pragma unroll
      for (int t = 0; t < SC; ++t) {
        const f32x4 v = (cmask >> t) & 1u ? conv_unpack(cu[t], g.cv.scale, g.cv.shift) : zero4;
        *reinterpret_cast<f32x4*>(Bs + (tid % KP) * (GN + 8) + 4 * (tid / KP + GB * t)) = v;
      }
    }
    if (GATE != TRL_ACT_NONE && pGate) {             // (a problem of a mixed launch may come without a gate)
#pragma unroll
      for (int t = 0; t < SA; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) ra[t][j] *= dact_from_out(GATE, rg[t][j]);
    }
    if (want_colsum) {
#pragma unroll
      for (int t = 0; t < SA; ++t) csum += ra[t];
    }
    if (CONV != 1) panel_stash<!TA, GM, KP>(As, tid, ra);
    if (CONV != 2) panel_stash<TB, GN, KP>(Bs, tid, rb);
  };
  // the MFMAs of 8-k group q of the panel in buffer `buf`: TMB + TMB operand reads feed 4 TMB^2 MFMAs
  auto group = [&](int buf, int q) {
    const float* As = lds + buf * (AF + BF);
    const float* Bs = As + AF;
    f32x4 av[TMB], bv[TMB];
#pragma unroll
    for (int a = 0; a < TMB; ++a) av[a] = panel_operand<!TA, GM, KP>(As, 32 * (TMB * wm + a) + i, q, hi);
#pragma unroll
    for (int b = 0; b < TMB; ++b) bv[b] = panel_operand<TB, GN, KP>(Bs, 32 * (TMB * wn + b) + i, q, hi);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int a = 0; a < TMB; ++a)
#pragma unroll
        for (int b = 0; b < TMB; ++b) acc[a][b] = mfma32(av[a][r], bv[b][r], acc[a][b]);
  };

  GCLK(0)
  const int np = (k_hi - k_lo + KP - 1) / KP;      // panels of this workgroup's reduction range
  // Dense interior tiles walk their WHOLE panels in a loop of their own that contains nothing but unconditional 16-byte
  // loads, LDS traffic and MFMAs.  (With the predicated edge loads in the same loop body the compiler shares registers
  // between their temporaries and the MFMA operands and, to keep that safe, waits for ALL outstanding loads at the top
  // of every panel -- the prefetch distance shrinks from a whole panel to half of one.)  Edge tiles, the reduction's
  // ragged tail and the implicit-operand modes take the general loop below.
  const bool fast_ok = CONV == 0 && a_whole && b_whole;
  const int nw = fast_ok ? (k_hi - k_lo) / KP : 0;
  if (CONV == 0 && nw > 0) {
    auto fetch_whole = [&](int k0) {
      panel_fetch_fast<!TA, GM, KP>(pA, g_lda, m0, k0, tid, ra);
      if (GATE != TRL_ACT_NONE && pGate) panel_fetch_fast<!TA, GM, KP>(pGate, g_lda, m0, k0, tid, rg);
      panel_fetch_fast<TB, GN, KP>(pB, g_ldb, n0, k0, tid, rb);
    };
    fetch_whole(k_lo);
    GCLK(1)
    stash(0);
    if (nw > 1) fetch_whole(k_lo + KP);
    __syncthreads();
    GCLK(2)
    for (int p = 0; p < nw; ++p) {
      const int cur = p & 1;
      // half of the panel's MFMAs; panel p + 1: registers -> the idle buffer (last read before the previous barrier);
      // panel p + 2: loads in flight across a whole panel; the other half
#pragma unroll
      for (int q = 0; q < NQ / 2; ++q) group(cur, q);
      if (p + 1 < nw) stash(cur ^ 1);
      if (p + 2 < nw) fetch_whole(k_lo + (p + 2) * KP);
#pragma unroll
      for (int q = NQ / 2; q < NQ; ++q) group(cur, q);
      __syncthreads();
      GCLK(p == 0 ? 3 : 5)
    }
  }
  if (nw < np) {                                   // general loop: panels [nw, np), same pipeline, edge-capable loads
    fetch(k_lo + nw * KP);
    stash(nw & 1);
    if (nw + 1 < np) fetch(k_lo + (nw + 1) * KP);
    __syncthreads();
    for (int p = nw; p < np; ++p) {
      const int k0 = k_lo + p * KP, cur = p & 1;
      // 8-k groups that hold data (the tail of the last panel is zero filled: skipping it changes no sum); rounded to the
      // 32-k rounds of the one-buffer loop so that edge panels walk the same groups as before
      const int nq = min(NQ, ((min(KP, k_hi - k0) + 31) >> 5) * 4);
      const int half = min(nq, NQ / 2);
      for (int q = 0; q < half; ++q) group(cur, q);
      if (p + 1 < np) stash(cur ^ 1);
      if (p + 2 < np) fetch(k0 + 2 * KP);
      for (int q = half; q < nq; ++q) group(cur, q);
      __syncthreads();
    }
  }
  // ---- epilogue: lane (i, hi) owns column n of rows rowmap(0..15, hi) of each of the wave's blocks ----
#pragma unroll
  for (int ab = 0; ab < TMB; ++ab)
#pragma unroll
  for (int bb = 0; bb < TMB; ++bb) {
    const int n = n0 + 32 * (TMB * wn + bb) + i, mb = m0 + 32 * (TMB * wm + ab) + 4 * hi;
    if (n < g_N) {
      const float bias = pBias ? pBias[n] : 0.0f;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[ab][bb][r] + bias;
      if (g.act == TRL_ACT_TANH) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = trl_tanh(v[r]);
      } else if (g.act == TRL_ACT_RELU) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.0f);
      }
      float* cp = C + (size_t)mb * g_ldc + n;
      if (CONV == 3 && g.chw_p) {
        // the trunk's last layer: the FC layers above read (b, c, oy, ox)-flattened features, so the tile is stored in that
        // order here (a lane's 4 consecutive rows are 16 contiguous bytes) instead of being transposed by a launch of its own
        const uint32_t P = (uint32_t)g.chw_p;
#pragma unroll
        for (int q = 0; q < 4; ++q) {                // rows mb + 8 q .. + 3: one position run of one sample, mostly
          const uint32_t m = (uint32_t)(mb + 8 * q);
          const uint32_t b = fastdiv(m, P, g.cv.hw_magic, g.cv.hw_shift), p = m - b * P;
          float* dst = C + ((size_t)b * g_N + n) * P + p;
          if ((int)m + 3 < g_M && p + 3 < P) {
            f32x4 pk = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            __builtin_memcpy(dst, &pk, 16);          // (4-byte aligned: P is odd for the 7 x 7 maps)
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const uint32_t mj = m + j;
              if ((int)mj < g_M) {
                const uint32_t bj = fastdiv(mj, P, g.cv.hw_magic, g.cv.hw_shift);
                C[((size_t)bj * g_N + n) * P + (mj - bj * P)] = v[4 * q + j];
              }
            }
          }
        }
      } else if (m0 + GM <= g_M) {
#pragma unroll
        for (int r = 0; r < 16; ++r) cp[(size_t)((r & 3) + 8 * (r >> 2)) * g_ldc] = v[r];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (mb + (r & 3) + 8 * (r >> 2) < g_M) cp[(size_t)((r & 3) + 8 * (r >> 2)) * g_ldc] = v[r];
      }
    }
  }
  GCLK(6)
  if (want_colsum) {
    // thread (tid / TPR, tid % TPR) holds partials of columns 4 * (tid % TPR) .. + 3: fold the row groups in order
    constexpr int TPR = GM / 4, GROUPS = 256 / TPR;
    float* s = lds;                                 // reuse (all MFMA reads are behind the last barrier)
    *reinterpret_cast<f32x4*>(s + (tid / TPR) * GM + 4 * (tid % TPR)) = csum;
    __syncthreads();
    if (tid < GM && m0 + tid < g_M) {
      float a = 0.0f;
#pragma unroll
      for (int w = 0; w < GROUPS; ++w) a += s[w * GM + tid];
      pColsum[(size_t)blockIdx.z * g_M + m0 + tid] = a;
    }
  }
}

// fixed-order fold of split partials: out[e] = sum_s part[s][e]; a second segment (the bias gradient) rides along.
// A workgroup owns 64 outputs; its W waves (4, or 16 when there are many splits and few outputs -- a conv layer's
// weight gradient is a few thousand floats in hundreds of partials, and 4 waves x 65 workgroups walked them for 32 us)
// each sum every W-th split, then the W slices are added in order.
#define FOLD_OUT 64
#define FOLD_MAX_WAVES 16
struct FoldGroup { const float* part; float* out; const float* part2; float* out2; };
struct FoldDev {
  int n, n2, splits;
  const float* bias; int n_cols, act;             // split-K forward: the epilogue the GEMM skipped (n_cols > 0)
  const float* bias_grp[GEMM_MAX_GROUPS];          //   (grouped: the bias of problem blockIdx.y, when gridDim.y > 1)
  int perm_c, perm_khw;                            // conv weight gradient computed in (i, j, c) column order
  FoldGroup grp[GEMM_MAX_GROUPS];                  // problem blockIdx.y
};
__global__ __launch_bounds__(64 * FOLD_MAX_WAVES) void fold_partials_kernel(FoldDev f) {
  __shared__ float sl[FOLD_MAX_WAVES][FOLD_OUT];
  const FoldGroup& q = f.grp[blockIdx.y];
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6, waves = blockDim.x >> 6;
  int e = blockIdx.x * FOLD_OUT + lane;
  const bool second = e >= f.n;                    // per lane: a workgroup may straddle the two segments
  const float* p = second ? q.part2 : q.part;
  const int nn = second ? f.n2 : f.n, ee = second ? e - f.n : e;
  float a = 0.0f;
  if (ee < nn) {
    int s = slice;
    for (; s + 3 * waves < f.splits; s += 4 * waves) {           // 4 loads in flight, added in split order
      const float x0 = p[(size_t)s * nn + ee], x1 = p[(size_t)(s + waves) * nn + ee];
      const float x2 = p[(size_t)(s + 2 * waves) * nn + ee], x3 = p[(size_t)(s + 3 * waves) * nn + ee];
      a += x0; a += x1; a += x2; a += x3;
    }
    for (; s < f.splits; s += waves) a += p[(size_t)s * nn + ee];
  }
  sl[slice][lane] = a;
  __syncthreads();
  if (slice == 0 && ee < nn) {
    float v = (sl[0][lane] + sl[1][lane]) + (sl[2][lane] + sl[3][lane]);
    for (int w = 4; w < waves; w += 4) v += (sl[w][lane] + sl[w + 1][lane]) + (sl[w + 2][lane] + sl[w + 3][lane]);
    if (!second && f.n_cols > 0) {
      const float* bias = gridDim.y > 1 ? f.bias_grp[blockIdx.y] : f.bias;
      if (bias) v += bias[ee % f.n_cols];
      if (f.act == TRL_ACT_TANH) v = trl_tanh(v);
      else if (f.act == TRL_ACT_RELU) v = fmaxf(v, 0.0f);
    }
    int eo = ee;
    if (!second && f.perm_c > 0) {                 // store as (c, i, j)
      const int K = f.perm_c * f.perm_khw, row = ee / K, kp = ee - row * K, ij = kp / f.perm_c, c = kp - ij * f.perm_c;
      eo = row * K + c * f.perm_khw + ij;
    }
    (second ? q.out2 : q.out)[eo] = v;
  }
}

// ---- fold scope: the weight-gradient folds of SEVERAL layers as one launch ----
// Between trl_fold_scope_begin() and trl_fold_scope_end(stream) (host state of the calling thread) every single-problem
// weight-gradient fold is RECORDED instead of launched -- the entry points that produce split partials (dense, implicit conv,
// direct first conv layer) need no second form -- and `end` folds them all in one launch: same arithmetic, same summation
// order, same wave count per fold as the launches it replaces (bit-identical), one dependent launch of ~5 us instead of one
// per layer.  The caller keeps the partials of different layers in different workspace regions and reads no gradient in
// between.
#define FOLD_SCOPE_MAX 8
struct FoldScopeEntry { int n, n2, splits, perm_c, perm_khw, waves, first_block; const float* part; float* out;
                        const float* part2; float* out2; };
struct FoldScopeDev { int count; FoldScopeEntry e[FOLD_SCOPE_MAX]; };
static thread_local struct { bool on; int blocks; FoldScopeDev d; } g_fold_scope;
__global__ __launch_bounds__(64 * FOLD_MAX_WAVES) void fold_scope_kernel(FoldScopeDev f) {
  __shared__ float sl[FOLD_MAX_WAVES][FOLD_OUT];
  int k = 0;
  while (k + 1 < f.count && (int)blockIdx.x >= f.e[k + 1].first_block) ++k;
  const FoldScopeEntry q = f.e[k];
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6, waves = q.waves;
  const int e = ((int)blockIdx.x - q.first_block) * FOLD_OUT + lane;
  const bool second = e >= q.n;
  const float* p = second ? q.part2 : q.part;
  const int nn = second ? q.n2 : q.n, ee = second ? e - q.n : e;
  float a = 0.0f;
  if (slice < waves && ee < nn) {                  // (fold_partials_kernel's walk, with this fold's own wave count)
    int s = slice;
    for (; s + 3 * waves < q.splits; s += 4 * waves) {
      const float x0 = p[(size_t)s * nn + ee], x1 = p[(size_t)(s + waves) * nn + ee];
      const float x2 = p[(size_t)(s + 2 * waves) * nn + ee], x3 = p[(size_t)(s + 3 * waves) * nn + ee];
      a += x0; a += x1; a += x2; a += x3;
    }
    for (; s < q.splits; s += waves) a += p[(size_t)s * nn + ee];
  }
  if (slice < waves) sl[slice][lane] = a;
  __syncthreads();
  if (slice == 0 && ee < nn) {
    float v = (sl[0][lane] + sl[1][lane]) + (sl[2][lane] + sl[3][lane]);
    for (int w = 4; w < waves; w += 4) v += (sl[w][lane] + sl[w + 1][lane]) + (sl[w + 2][lane] + sl[w + 3][lane]);
    int eo = ee;
    if (!second && q.perm_c > 0) {
      const int K = q.perm_c * q.perm_khw, row = ee / K, kp = ee - row * K, ij = kp / q.perm_c, c = kp - ij * q.perm_c;
      eo = row * K + c * q.perm_khw + ij;
    }
    (second ? q.out2 : q.out)[eo] = v;
  }
}
extern "C" int trl_fold_scope_begin(void) {
  TRL_REQUIRE(!g_fold_scope.on, "fold scope: already open on this thread");
  g_fold_scope.on = true; g_fold_scope.blocks = 0; g_fold_scope.d.count = 0;
  return TRL_OK;
}
extern "C" int trl_fold_scope_end(void* stream) {
  TRL_REQUIRE(g_fold_scope.on, "fold scope: not open");
  g_fold_scope.on = false;
  if (g_fold_scope.d.count == 0) return TRL_OK;
  hipLaunchKernelGGL(fold_scope_kernel, dim3(g_fold_scope.blocks), dim3(64 * FOLD_MAX_WAVES), 0, (hipStream_t)stream, g_fold_scope.d);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

static int launch_fold(FoldDev f, int groups, hipStream_t s, bool immediate = false) {
  const int blocks = trl_ceil_div((int64_t)f.n + f.n2, FOLD_OUT);
  // many splits behind few workgroups: 16 waves share them (latency-bound walk); otherwise 4
  const int waves = (f.splits >= 32 && (int64_t)blocks * groups < 2048) ? FOLD_MAX_WAVES : 4;
  if (g_fold_scope.on && !immediate && groups == 1 && f.n_cols == 0 && g_fold_scope.d.count < FOLD_SCOPE_MAX) {
    const FoldGroup& q = f.grp[0];
    g_fold_scope.d.e[g_fold_scope.d.count++] = FoldScopeEntry{f.n, f.n2, f.splits, f.perm_c, f.perm_khw, waves, g_fold_scope.blocks,
                                                              q.part, q.out, q.part2, q.out2};
    g_fold_scope.blocks += blocks;
    return TRL_OK;
  }
  hipLaunchKernelGGL(fold_partials_kernel, dim3(blocks, groups), dim3(64 * waves), 0, s, f);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// Several independent folds in ONE launch: the weight (and bias) gradients of every layer of a backward pass leave
// their partials in place (trl_linear_bwd_weight_partials_group_f32) and are folded together at the end -- a 256-wide
// MLP's six per-layer folds are ~5 us of dependent launch each for a few hundred KB of work.
#define FOLD_MULTI_MAX 32
struct FoldMulti {
  int count;
  int first_block[FOLD_MULTI_MAX + 1];             // entry k owns blocks [first_block[k], first_block[k + 1])
  int n[FOLD_MULTI_MAX], splits[FOLD_MULTI_MAX];
  const float* part[FOLD_MULTI_MAX]; float* out[FOLD_MULTI_MAX];
};
__global__ __launch_bounds__(256) void fold_multi_kernel(FoldMulti f) {
  __shared__ float sl[4][FOLD_OUT];
  int k = 0;
  while (k + 1 < f.count && (int)blockIdx.x >= f.first_block[k + 1]) ++k;
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int e = (blockIdx.x - f.first_block[k]) * FOLD_OUT + lane, nn = f.n[k], splits = f.splits[k];
  const float* p = f.part[k];
  float a = 0.0f;
  if (e < nn) {
    int s = slice;
    for (; s + 12 < splits; s += 16) {
      const float x0 = p[(size_t)s * nn + e], x1 = p[(size_t)(s + 4) * nn + e];
      const float x2 = p[(size_t)(s + 8) * nn + e], x3 = p[(size_t)(s + 12) * nn + e];
      a += x0; a += x1; a += x2; a += x3;
    }
    for (; s < splits; s += 4) a += p[(size_t)s * nn + e];
  }
  sl[slice][lane] = a;
  __syncthreads();
  if (slice == 0 && e < nn) f.out[k][e] = (sl[0][lane] + sl[1][lane]) + (sl[2][lane] + sl[3][lane]);   // = fold_partials_kernel
}
extern "C" int trl_fold_partials_multi_f32(int count, const float* const* part, float* const* out, const int* n,
                                           const int* splits, void* stream) {
  TRL_REQUIRE(count >= 1 && count <= FOLD_MULTI_MAX, "fold_multi: 1..32 folds per launch");
  TRL_REQUIRE(part && out && n && splits, "null pointer");
  FoldMulti f{};
  f.count = count;
  for (int k = 0; k < count; ++k) {
    TRL_REQUIRE(part[k] && out[k] && n[k] > 0 && splits[k] >= 1, "fold_multi: bad entry");
    f.part[k] = part[k]; f.out[k] = out[k]; f.n[k] = n[k]; f.splits[k] = splits[k];
    f.first_block[k + 1] = f.first_block[k] + trl_ceil_div(n[k], FOLD_OUT);
  }
  hipLaunchKernelGGL(fold_multi_kernel, dim3(f.first_block[count]), dim3(256), 0, (hipStream_t)stream, f);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

int trl_fold_partials(const float* part, float* out, int n, const float* part2, float* out2, int n2, int splits,
                      hipStream_t stream) {
  FoldDev f{};
  f.n = n; f.n2 = n2; f.splits = splits;
  f.grp[0] = FoldGroup{part, out, part2, out2};
  return launch_fold(f, 1, stream);
}

template <bool TA, bool TB, int GATE, int CONV, int WM, int TMB = 1, int KPX = 0>
static int launch_gemm_tile(GemmDev g, int splits, hipStream_t s) {
  constexpr int KP = KPX ? KPX : gemm_panel<TMB>();
  constexpr int GM = 32 * WM * TMB, GN = 32 * (4 / WM) * TMB;
  const int tiles_lds = 2 * (int)sizeof(float) * (tile_floats<!TA, GM, KP>() + tile_floats<TB, GN, KP>());   // two panel buffers
  const int lds = tiles_lds + (CONV == 1 ? g.K : 0);              // CONV 1: + the tap offset table (K / 4 dwords)
  TRL_REQUIRE(lds <= 160 * 1024, "reduction too long for the implicit first-layer kernel");
  static int attr_lds = 0;
  if (lds > attr_lds) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_f32_kernel<TA, TB, GATE, CONV, WM, TMB, KPX>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { trl_set_error("gemm: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_lds = lds;
  }
  g.tiles_n = trl_ceil_div(g.N, GN);
  g.tiles = g.tiles_n * trl_ceil_div(g.M, GM);
  hipLaunchKernelGGL((gemm_f32_kernel<TA, TB, GATE, CONV, WM, TMB, KPX>), dim3(g.tiles, std::max(1, g.groups), splits), dim3(256), lds, s, g);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// waves along M of the C tile: narrow outputs get the 128 x 32 tile, few-row ones the 32 x 128 tile
static int tile_wm(int M, int N) { return N <= 32 ? 4 : (M <= 32 ? 1 : 2); }
// Large dense products take the 128 x 128 tile (2 x 2 accumulator blocks per wave: half the LDS reads per MFMA, 64 MFMAs
// per wave and barrier) once it still leaves two rounds of workgroups for 256 CUs x 2 resident workgroups; everything
// smaller keeps the 64 x 64 tile, whose count is what fills the chip.  TRL_GEMM_TILE=64 / 128 pins either (development).
static bool tile_128(const GemmDev& g, int splits) {
  static const int pin = [] { const char* e = getenv("TRL_GEMM_TILE"); return e ? atoi(e) : 0; }();
  if (g.hetero || g.M < 128 || g.N < 128) return false;
  if (pin == 64) return false;
  if (pin == 128 || g.prefer128) return true;
  const int64_t tiles = (int64_t)trl_ceil_div(g.M, 128) * trl_ceil_div(g.N, 128) * std::max(1, g.groups) * splits;
  return tiles >= 1024;
}

// Short reductions on many tiles (<= 8 panels of 64 each): the 32-wide panel -- half the LDS, twice the resident
// workgroups (64 x 64 tile: four per CU instead of two; the 128 x 32 tile of narrow conv layers: three instead of ONE).
// Dense products want >= 3 rounds of 256 workgroups before it pays, the implicit forward on the 128 x 32 tile >= 2.  TRL_GEMM_KP=64 / 32 pins either (development).
static bool short_reduction(const GemmDev& g, int splits, int gm, int gn, int conv) {
  static const int pin = [] { const char* e = getenv("TRL_GEMM_KP"); return e ? atoi(e) : 0; }();
  if (g.hetero || pin == 64) return false;
  const int64_t wgs = (int64_t)trl_ceil_div(g.M, gm) * trl_ceil_div(g.N, gn) * std::max(1, g.groups) * splits;
  const int red = splits > 1 ? g.split_len : g.K;
  if (pin == 32) return true;
  if (conv) {
    // implicit forward: the 128 x 32 tile holds ONE workgroup per CU with 64-deep panels, three with 32-deep ones -- it pays
    // as soon as the grid is more than one round of 256; the 64 x 64 tile (two per CU, four with 32-deep panels) only with
    // pre-ordered weights (dense B loads) and more than one round of 512
    if (gm == 128) return wgs > 256 && red <= 512;
    return g.b_perm && wgs > 512 && wgs <= 1024 && red <= 512;
  }
  return wgs >= 768 && red <= 512;
}

template <bool TA, bool TB, int GATE, int CONV>
static int launch_gemm_gate(const GemmDev& g, int splits, hipStream_t s) {
  switch (tile_wm(g.M, g.N)) {
    case 4:
      if constexpr (CONV == 3) { if (short_reduction(g, splits, 128, 32, CONV)) return launch_gemm_tile<TA, TB, GATE, CONV, 4, 1, 32>(g, splits, s); }
      return launch_gemm_tile<TA, TB, GATE, CONV, 4>(g, splits, s);
    case 1:  return launch_gemm_tile<TA, TB, GATE, CONV, 1>(g, splits, s);
    default:
      if constexpr (CONV == 0) { if (tile_128(g, splits)) return launch_gemm_tile<TA, TB, GATE, CONV, 2, 2>(g, splits, s); }
      // (the implicit forward on 64 x 64 tiles decodes its taps once per panel and, without pre-ordered weights, gathers B
      // with strided 4-byte loads: with twice the panels that lost, 56.6 vs 42 us)
      if constexpr (CONV == 0 || CONV == 3) {
        if (short_reduction(g, splits, 64, 64, CONV)) return launch_gemm_tile<TA, TB, GATE, CONV, 2, 1, 32>(g, splits, s);
      }
      return launch_gemm_tile<TA, TB, GATE, CONV, 2>(g, splits, s);
  }
}

template <bool TA, bool TB, int CONV = 0>
static int launch_gemm(const GemmDev& g, int splits, hipStream_t s) {
  const int gate = (g.groups > 1 ? g.grp[0].a_gate : g.a_gate) ? g.gate_act : TRL_ACT_NONE;
  if (gate == TRL_ACT_TANH) return launch_gemm_gate<TA, TB, TRL_ACT_TANH, CONV>(g, splits, s);
  if (gate == TRL_ACT_RELU) return launch_gemm_gate<TA, TB, TRL_ACT_RELU, CONV>(g, splits, s);
  return launch_gemm_gate<TA, TB, TRL_ACT_NONE, CONV>(g, splits, s);
}

// Reduction rows per blockIdx.z of the weight-gradient GEMM: never below 256 (two panels), and long enough that
// the grid stays near 1024 workgroups when M is huge (conv layers: M = B * Ho * Wo).
static int bw_split_len(int M, int K, int N) {
  const int wm = tile_wm(N, K);                    // the weight-gradient GEMM is (N x K) with reduction M
  const int tiles = trl_ceil_div(N, 32 * wm) * trl_ceil_div(K, 32 * (4 / wm));
  // (>= 320 tiles already fill the chip: no split, and then no partials and no fold either -- the 512 x 3136 layer of the
  // conv nets wrote and folded 2 x 6.4 MB for nothing)
  // (the 32 x 128 / 128 x 32 tiles hold 87 KB of panel buffers: ONE workgroup per CU, so the grid aims at one round of 256
  // -- 324 workgroups were a full round plus a quarter-full one, 26 us for conv 2 of the Atari trunk)
  const int want = tiles >= 320 ? 1 : std::max(1, (wm == 2 ? 1024 : 256) / tiles);
  const int len = trl_ceil_div(trl_ceil_div(M, want), KC) * KC;
  return std::max(256, len);
}

static int linear_fwd_impl(int G, const float* const* x, const float* const* w, const float* const* bias, float* const* y,
                           int M, int K, int N, int act, hipStream_t stream) {
  TRL_REQUIRE(G >= 1 && G <= GEMM_MAX_GROUPS, "1..12 problems per grouped launch");
  TRL_REQUIRE(M >= 0 && K > 0 && N > 0, "bad sizes");
  if (M == 0) return TRL_OK;
  TRL_REQUIRE(act == TRL_ACT_TANH || act == TRL_ACT_RELU || act == TRL_ACT_NONE, "unknown activation");
  GemmDev g{};
  g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N; g.act = act; g.gate_act = TRL_ACT_NONE; g.split_len = K;
  g.groups = G;
  for (int i = 0; i < G; ++i) {
    TRL_REQUIRE(x[i] && w[i] && y[i], "null pointer");
    g.grp[i] = GemmGroup{x[i], w[i], y[i], bias ? bias[i] : nullptr, nullptr, nullptr};
  }
  g.A = x[0]; g.B = w[0]; g.C = y[0]; g.bias = bias ? bias[0] : nullptr;
  return launch_gemm<false, true>(g, 1, stream);
}
extern "C" int trl_linear_fwd_f32(const float* x, const float* w, const float* bias, float* y, int M, int K, int N,
                                  int act, void* stream) {
  return linear_fwd_impl(1, &x, &w, &bias, &y, M, K, N, act, (hipStream_t)stream);
}
extern "C" int trl_linear_fwd_group_f32(int G, const float* const* x, const float* const* w, const float* const* bias,
                                        float* const* y, int M, int K, int N, int act, void* stream) {
  TRL_REQUIRE(x && w && y, "null pointer array");
  return linear_fwd_impl(G, x, w, bias, y, M, K, N, act, (hipStream_t)stream);
}

// Split-K forward for few-row layers with a long reduction (the conv nets' first FC layer: 512 x 3136 -> 512 is
// 64 C tiles for 256 CUs): up to 8 reduction slices write partial products, the fold adds bias and activation.
// Few output tiles and a long reduction (the conv nets' first FC layer: 512 x 3136 -> 512, 64 tiles of 64 x 64): with
// 128 x 128 tiles and 8 reduction slices a pair of such layers is exactly one workgroup per CU, each on the more
// efficient tile (5 slices of 64 x 64 tiles were 640 workgroups for 512 resident slots: 44.6 + 8 us per pair).
// (One such layer alone -- the collection pass -- would be 128 workgroups: it keeps the 64 x 64 tiles.)
static bool fwd_use_128(int M, int K, int N, int G) {
  static const bool off = [] { const char* e = getenv("TRL_FWD_SPLIT128"); return e && atoi(e) == 0; }();   // (development A/B)
  return !off && G >= 2 && M >= 128 && N >= 128 && K >= 1024 && trl_ceil_div(M, 64) * trl_ceil_div(N, 64) <= 96;
}
static int fwd_split_len(int M, int K, int N, int G = 1) {
  if (fwd_use_128(M, K, N, G)) {
    const int t128 = trl_ceil_div(M, 128) * trl_ceil_div(N, 128);
    const int target = std::max(1, std::min(8, trl_ceil_div(192, t128)));
    return trl_ceil_div(trl_ceil_div(K, target), KC) * KC;
  }
  const int wm = tile_wm(M, N);
  const int tiles = trl_ceil_div(M, 32 * wm) * trl_ceil_div(N, 32 * (4 / wm));
  // (a handful of tiles -- the 6-wide DQN head on 512 rows is FOUR workgroups walking K = 512 panel by panel, 17 us --
  // splits from 4 panels on)
  if (tiles >= 192 || K < (tiles <= 16 ? 4 : 8) * KC) return K;
  const int target = std::min(8, trl_ceil_div(384, tiles));
  return trl_ceil_div(trl_ceil_div(K, target), KC) * KC;
}
extern "C" int trl_linear_fwd_workspace(int M, int K, int N) {
  if (M <= 0 || K <= 0 || N <= 0) return 0;
  const int splits = std::max(trl_ceil_div(K, fwd_split_len(M, K, N, 1)), trl_ceil_div(K, fwd_split_len(M, K, N, 2)));   // either policy
  return splits > 1 ? splits * M * N : 0;
}
extern "C" int trl_linear_fwd_splitk_f32(const float* x, const float* w, const float* bias, float* y, int M, int K, int N,
                                         int act, float* workspace, void* stream) {
  TRL_REQUIRE(M >= 0 && K > 0 && N > 0, "bad sizes");
  if (M == 0) return TRL_OK;
  const int split_len = fwd_split_len(M, K, N);
  const int splits = trl_ceil_div(K, split_len);
  if (splits <= 1) return trl_linear_fwd_f32(x, w, bias, y, M, K, N, act, stream);
  TRL_REQUIRE(x && w && y && workspace, "null pointer");
  TRL_REQUIRE(act == TRL_ACT_TANH || act == TRL_ACT_RELU || act == TRL_ACT_NONE, "unknown activation");
  GemmDev g{};
  g.A = x; g.B = w; g.C = workspace; g.bias = nullptr; g.a_gate = nullptr; g.M = M; g.N = N; g.K = K;
  g.lda = K; g.ldb = K; g.ldc = N; g.act = TRL_ACT_NONE; g.gate_act = TRL_ACT_NONE; g.split_len = split_len; g.colsum = nullptr;
  int rc = launch_gemm<false, true>(g, splits, (hipStream_t)stream);
  if (rc) return rc;
  FoldDev f{};
  f.n = M * N; f.n2 = 0; f.splits = splits; f.bias = bias; f.n_cols = N; f.act = act;
  f.grp[0] = FoldGroup{workspace, y, nullptr, nullptr};
  return launch_fold(f, 1, (hipStream_t)stream);
}

// the same for G same-shaped layers (one launch of split GEMMs + one fold): workspace = G x trl_linear_fwd_workspace floats
extern "C" int trl_linear_fwd_splitk_group_f32(int G, const float* const* x, const float* const* w, const float* const* bias,
                                               float* const* y, int M, int K, int N, int act, float* workspace,
                                               void* stream) {
  TRL_REQUIRE(G >= 1 && G <= GEMM_MAX_GROUPS, "1..12 problems per grouped launch");
  TRL_REQUIRE(M >= 0 && K > 0 && N > 0 && x && w && y, "bad sizes / null pointer array");
  if (M == 0) return TRL_OK;
  const int split_len = fwd_split_len(M, K, N, G);
  const int splits = trl_ceil_div(K, split_len);
  if (splits <= 1) return linear_fwd_impl(G, x, w, bias, y, M, K, N, act, (hipStream_t)stream);
  TRL_REQUIRE(workspace, "null workspace");
  TRL_REQUIRE(act == TRL_ACT_TANH || act == TRL_ACT_RELU || act == TRL_ACT_NONE, "unknown activation");
  GemmDev g{};
  g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N; g.act = TRL_ACT_NONE; g.gate_act = TRL_ACT_NONE;
  g.split_len = split_len; g.groups = G;
  g.prefer128 = fwd_use_128(M, K, N, G) ? 1 : 0;
  FoldDev f{};
  f.n = M * N; f.n2 = 0; f.splits = splits; f.n_cols = N; f.act = act;
  const float* bias0 = bias ? bias[0] : nullptr;
  for (int i = 0; i < G; ++i) {
    TRL_REQUIRE(x[i] && w[i] && y[i], "null pointer");
    float* part = workspace + (size_t)i * splits * M * N;
    g.grp[i] = GemmGroup{x[i], w[i], part, nullptr, nullptr, nullptr};
    f.grp[i] = FoldGroup{part, y[i], nullptr, nullptr};
    f.bias_grp[i] = bias ? bias[i] : nullptr;
  }
  f.bias = bias0;
  g.A = x[0]; g.B = w[0]; g.C = g.grp[0].C;
  int rc = launch_gemm<false, true>(g, splits, (hipStream_t)stream);
  if (rc) return rc;
  return launch_fold(f, G, (hipStream_t)stream);
}

static int linear_bwd_input_impl(int G, const float* const* dy, const float* const* y_gate, int gate_act,
                                 const float* const* w, float* const* dx, int M, int K, int N, hipStream_t stream) {
  TRL_REQUIRE(G >= 1 && G <= GEMM_MAX_GROUPS, "1..12 problems per grouped launch");
  TRL_REQUIRE(M >= 0 && K > 0 && N > 0, "bad sizes");
  if (M == 0) return TRL_OK;
  GemmDev g{};
  g.gate_act = gate_act; g.M = M; g.N = K; g.K = N; g.lda = N; g.ldb = K; g.ldc = K; g.act = TRL_ACT_NONE; g.split_len = N;
  g.groups = G;
  const bool gated = y_gate && y_gate[0];
  for (int i = 0; i < G; ++i) {
    TRL_REQUIRE(dy[i] && w[i] && dx[i], "null pointer");
    TRL_REQUIRE(!gated || y_gate[i], "either every problem of a group is gated or none");
    g.grp[i] = GemmGroup{dy[i], w[i], dx[i], nullptr, gated ? y_gate[i] : nullptr, nullptr};
  }
  g.A = dy[0]; g.B = w[0]; g.C = dx[0]; g.a_gate = g.grp[0].a_gate;
  return launch_gemm<false, false>(g, 1, stream);
}
extern "C" int trl_linear_bwd_input_f32(const float* dy, const float* y_gate, int gate_act, const float* w, float* dx,
                                        int M, int K, int N, void* stream) {
  return linear_bwd_input_impl(1, &dy, &y_gate, gate_act, &w, &dx, M, K, N, (hipStream_t)stream);
}
// Few output tiles behind a long reduction (a wide head's input gradient: QR-DQN's 512 x 1200 head on 512 rows is 64 tiles
// walking 1200 outputs each -- 64 workgroups on 256 CUs, 31.8 us for 1.26 GFLOP): the reduction is split over slices like
// the forward's (trl_linear_fwd_splitk_f32), partial dX in the workspace, fixed-order fold.
static int bwdin_split_len(int M, int K, int N) {
  const int tiles = trl_ceil_div(M, 64) * trl_ceil_div(K, 64);
  if (tiles >= 128 || N < 8 * KC) return N;
  const int target = std::min(8, trl_ceil_div(256, tiles));
  return trl_ceil_div(trl_ceil_div(N, target), KC) * KC;
}
extern "C" int trl_linear_bwd_input_workspace(int M, int K, int N) {
  if (M <= 0 || K <= 0 || N <= 0) return 0;
  const int splits = trl_ceil_div(N, bwdin_split_len(M, K, N));
  return splits > 1 ? splits * M * K : 0;
}
extern "C" int trl_linear_bwd_input_splitk_f32(const float* dy, const float* y_gate, int gate_act, const float* w, float* dx,
                                               float* workspace, int M, int K, int N, void* stream) {
  TRL_REQUIRE(M >= 0 && K > 0 && N > 0, "bad sizes");
  if (M == 0) return TRL_OK;
  const int split_len = bwdin_split_len(M, K, N), splits = trl_ceil_div(N, split_len);
  if (splits <= 1) return linear_bwd_input_impl(1, &dy, &y_gate, gate_act, &w, &dx, M, K, N, (hipStream_t)stream);
  TRL_REQUIRE(dy && w && dx && workspace, "null pointer");
  GemmDev g{};
  g.gate_act = gate_act; g.M = M; g.N = K; g.K = N; g.lda = N; g.ldb = K; g.ldc = K; g.act = TRL_ACT_NONE; g.split_len = split_len;
  g.groups = 1;
  g.A = dy; g.B = w; g.C = workspace; g.a_gate = y_gate;
  g.grp[0] = GemmGroup{dy, w, workspace, nullptr, y_gate, nullptr};
  int rc = launch_gemm<false, false>(g, splits, (hipStream_t)stream);
  if (rc) return rc;
  FoldDev f{};
  f.n = M * K; f.n2 = 0; f.splits = splits; f.n_cols = 0;
  f.grp[0] = FoldGroup{workspace, dx, nullptr, nullptr};
  return launch_fold(f, 1, (hipStream_t)stream, /*immediate=*/true);   // (its consumer follows: never deferred by a fold scope)
}
extern "C" int trl_linear_bwd_input_group_f32(int G, const float* const* dy, const float* const* y_gate, int gate_act,
                                              const float* const* w, float* const* dx, int M, int K, int N, void* stream) {
  TRL_REQUIRE(dy && w && dx, "null pointer array");
  return linear_bwd_input_impl(G, dy, y_gate, gate_act, w, dx, M, K, N, (hipStream_t)stream);
}

extern "C" int trl_linear_bwd_weight_workspace(int M, int K, int N) {
  // floats of workspace for trl_linear_bwd_weight_f32 (split partials of dW and db)
  const int splits = M <= 0 ? 1 : trl_ceil_div(M, bw_split_len(M, K, N));
  return splits * (N * K + N);
}

template <int CONV>
static int bwd_weight_impl(int G, const float* const* dy, const float* const* y_gate, int gate_act, const float* const* x,
                           const ConvSrc* cv, float* const* dw, float* const* db, float* workspace, int M, int K, int N,
                           hipStream_t s, bool fold = true) {
  TRL_REQUIRE(G >= 1 && G <= GEMM_MAX_GROUPS, "1..12 problems per grouped launch");
  const int split_len = bw_split_len(M, K, N);
  const int splits = trl_ceil_div(M, split_len);
  const size_t per = (size_t)splits * ((size_t)N * K + N);       // workspace floats of one problem
  GemmDev g{};
  g.gate_act = gate_act; g.M = N; g.N = K; g.K = M; g.lda = N; g.ldb = K; g.ldc = K; g.act = TRL_ACT_NONE;
  g.split_len = split_len; g.groups = G;
  if (cv) g.cv = *cv;
  const bool gated = y_gate && y_gate[0];
  const bool want_db = db && db[0];
  const bool direct = splits == 1 && CONV == 0 && fold;            // (the conv fold also un-permutes the columns)
  FoldDev f{};
  f.n = N * K; f.n2 = want_db ? N : 0; f.splits = splits;
  f.perm_c = CONV == 4 ? cv->C : 0; f.perm_khw = CONV == 4 ? cv->kh * cv->kw : 0;
  for (int i = 0; i < G; ++i) {
    TRL_REQUIRE(dy[i] && (dw[i] || !fold) && (CONV != 0 || x[i]), "null pointer");
    TRL_REQUIRE(!gated || y_gate[i], "either every problem of a group is gated or none");
    TRL_REQUIRE(!want_db || db[i], "either every problem of a group wants db or none");
    float* part = workspace + i * per;
    float* cpart = want_db ? part + (size_t)splits * N * K : nullptr;
    if (direct) { part = dw[i]; cpart = want_db ? db[i] : nullptr; }   // one slice: the GEMM writes dW / db themselves
    g.grp[i] = GemmGroup{dy[i], CONV != 0 ? nullptr : x[i], part, nullptr, gated ? y_gate[i] : nullptr, cpart};
    f.grp[i] = FoldGroup{part, dw[i], cpart, want_db ? db[i] : nullptr};
  }
  g.A = g.grp[0].A; g.B = g.grp[0].B; g.C = g.grp[0].C; g.a_gate = g.grp[0].a_gate; g.colsum = g.grp[0].colsum;
  int rc = launch_gemm<true, false, CONV>(g, splits, s);
  if (rc || !fold || direct) return rc;
  return launch_fold(f, G, s);
}
extern "C" int trl_linear_bwd_weight_f32(const float* dy, const float* y_gate, int gate_act, const float* x, float* dw,
                                         float* db, float* workspace, int M, int K, int N, void* stream) {
  TRL_REQUIRE(M > 0 && K > 0 && N > 0, "bad sizes");
  TRL_REQUIRE(dy && x && dw && workspace, "null pointer");
  return bwd_weight_impl<0>(1, &dy, &y_gate, gate_act, &x, nullptr, &dw, &db, workspace, M, K, N, (hipStream_t)stream);
}
extern "C" int trl_linear_bwd_weight_group_f32(int G, const float* const* dy, const float* const* y_gate, int gate_act,
                                               const float* const* x, float* const* dw, float* const* db, float* workspace,
                                               int M, int K, int N, void* stream) {
  TRL_REQUIRE(M > 0 && K > 0 && N > 0, "bad sizes");
  TRL_REQUIRE(dy && x && dw && workspace, "null pointer array");
  return bwd_weight_impl<0>(G, dy, y_gate, gate_act, x, nullptr, dw, db, workspace, M, K, N, (hipStream_t)stream);
}
extern "C" int trl_linear_bwd_weight_splits(int M, int K, int N) {
  return M <= 0 ? 1 : trl_ceil_div(M, bw_split_len(M, K, N));
}
// The GEMM half only: problem i leaves S = trl_linear_bwd_weight_splits(M, K, N) partials of dW at
// workspace + i * S * (N * K + N) ([S][N * K]) followed, when want_db, by S partials of db ([S][N]); fold them with
// trl_fold_partials_multi_f32 (same summation order as the folding entry points: bit-identical results).
extern "C" int trl_linear_bwd_weight_partials_group_f32(int G, const float* const* dy, const float* const* y_gate,
                                                        int gate_act, const float* const* x, int want_db,
                                                        float* workspace, int M, int K, int N, void* stream) {
  TRL_REQUIRE(M > 0 && K > 0 && N > 0, "bad sizes");
  TRL_REQUIRE(dy && x && workspace && G >= 1 && G <= GEMM_MAX_GROUPS, "null pointer array / bad group count");
  float* none[GEMM_MAX_GROUPS] = {};
  float* some[GEMM_MAX_GROUPS];
  for (int i = 0; i < GEMM_MAX_GROUPS; ++i) some[i] = workspace;     // only "is a bias gradient wanted" is read
  return bwd_weight_impl<0>(G, dy, y_gate, gate_act, x, nullptr, none, want_db ? some : none, workspace, M, K, N,
                            (hipStream_t)stream, false);
}
// ---- weight gradients of SEVERAL layers (different K, N; same batch M) as one launch of split GEMMs ----
// 64 x 64 tiles for every problem (narrow layers compute padding: they are latency-, not MFMA-bound); splits chosen so
// that a problem contributes ~256 workgroups.  Partials are left in place for trl_fold_partials_multi_f32.
//
// SKINNY layers -- few inputs (the first layer of an MLP on 17- / 23-wide observations) or few outputs (a 1- / 12-wide
// head) against a wide other side -- take skinny_bwdw_kernel below instead: dW = dZ^T X is then a (wide x <= 32) outer
// product summed over the batch, and both operands of a v_mfma_f32_32x32x2_f32 step can be read from global memory
// DIRECTLY in operand layout (lane (i, hi) of wave w: wide[m + hi][32 w + i] -- 128-byte row pieces -- and
// narrow[m + hi][i], masked): no LDS, no barrier, no per-element edge code.  Through the generic kernel these layers are
// all edge tiles (predicated scalar loads: ~9 vector instructions per MFMA measured over a SAC update's launches) and the
// launch that carries the six of them took as long as the three 256 x 256 layers' (24.5 us for 0.2 GFLOP).
static bool skinny_layer(int K, int N) {
  return (K <= 31 && N >= 64 && N <= 256 && (N & 31) == 0) || (N <= 32 && K >= 64 && K <= 256 && (K & 31) == 0);
}
static int multi_split_len(int M, int K, int N) {
  if (skinny_layer(K, N))                                              // batch rows per workgroup: >= 128, ~64 slices
    return std::max(128, trl_ceil_div(trl_ceil_div(M, 64), KC) * KC);
  static const int target = [] { const char* e = getenv("TRL_BWW_TARGET"); return e ? atoi(e) : 256; }();   // (development)
  const int tiles = trl_ceil_div(N, 64) * trl_ceil_div(K, 64);
  const int want = std::max(1, target / tiles);
  return std::max(tiles <= 8 ? KC : 256, trl_ceil_div(trl_ceil_div(M, want), KC) * KC);   // (a handful of tiles: one panel per slice)
}
extern "C" int trl_linear_bwd_weight_multi_splits(int M, int K, int N) {
  return (M <= 0 || K <= 0 || N <= 0) ? 1 : trl_ceil_div(M, multi_split_len(M, K, N));
}

// One skinny layer per blockIdx.y, one batch slice per blockIdx.x.  `wide` is the operand with the many columns (dZ when
// the layer has few inputs -- wide_is_out -- else the layer input X), `narrow` the other one; the gate (act' through the
// layer's outputs) belongs to dZ, whichever side that is.  Wave w owns wide columns [32 w, 32 w + 32); its accumulator tile
// is (wide column) x (narrow column).  The bias gradient (column sums of dZ) rides along: as a column of ones appended to
// the narrow operand (wide_is_out), or as the per-lane running sum of the narrow operand (else).
struct SkinnyProb { const float* wide; const float* narrow; const float* gate; float* dw_part; float* db_part;
                    int wide_n, narrow_n, wide_is_out, split_len, splits; };
struct SkinnyDev { int M, gate_act; SkinnyProb p[GEMM_MAX_GROUPS]; };
template <int GATE>                                  // != NONE: some problem of the launch is gated (the others read a dummy)
__global__ __launch_bounds__(512) void skinny_bwdw_kernel(SkinnyDev g) {
  constexpr bool GATED = GATE != TRL_ACT_NONE;
  const SkinnyProb P = g.p[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, hi = lane >> 5;
  if ((int)blockIdx.x >= P.splits || 32 * wave >= P.wide_n) return;
  const int m_lo = blockIdx.x * P.split_len, m_hi = min(g.M, m_lo + P.split_len);
  const int wn = P.wide_n, nn = P.narrow_n;
  const bool gate_wide = P.gate && P.wide_is_out, gate_narrow = P.gate && !P.wide_is_out;
  const bool nin = i < nn;                           // this lane's narrow column exists
  const bool ones = P.wide_is_out && i == nn && P.db_part;   // ... or is the column of ones that collects db
  const float* wp = P.wide + 32 * wave + i;
  const float* np_ = P.narrow + (nin ? i : 0);
  // (no branch around a load -- the wait-count bookkeeping would have to assume it not taken: an ungated problem of a
  // gated launch reads its own wide operand as a dummy)
  const float* gp = P.gate ? (P.wide_is_out ? P.gate + 32 * wave + i : P.gate + (nin ? i : 0)) : wp;
  const size_t gld = (size_t)((P.wide_is_out || !P.gate) ? wn : nn);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  float nsum = 0.0f;                                 // !wide_is_out: running sum of this lane's dZ values (bias gradient)
  // 8 row pairs per round, one round in flight.  (A second round requested ahead of this one's MFMAs measured SLOWER,
  // 18-19 us against 16.4 for a SAC update's six skinny layers; so did the same direct-operand scheme for the square
  // 256 x 256 layers -- 34 us against 28.6 through LDS: 4-byte-per-lane loads move ~10 bytes per clock and CU here, so
  // this kernel is for layers whose generic tiles are all padding, not a replacement for the LDS-staged panels.)
  constexpr int U = 8;
  for (int m0 = m_lo; m0 < m_hi; m0 += 2 * U) {
    float av[U], bv[U], gv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = m0 + 2 * u + hi;
      const size_t mr = m < m_hi ? (size_t)m : (size_t)m_lo;           // (past the end: an in-range row, zeroed below)
      av[u] = wp[mr * wn];
      bv[u] = np_[mr * nn];
      if (GATED) gv[u] = gp[mr * gld];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = m0 + 2 * u + hi < m_hi;
      float a = ok ? av[u] : 0.0f, b = ok ? (nin ? bv[u] : (ones ? 1.0f : 0.0f)) : 0.0f;
      if (GATED) {
        const float d = dact_from_out(GATE, gv[u]);   // (compile-time activation: a run-time one is a branch chain per element)
        a *= gate_wide ? d : 1.0f;
        b *= gate_narrow ? d : 1.0f;
      }
      nsum += b;
      acc = mfma32(a, b, acc);
    }
  }
  // tile element (row = wide column 32 w + rowmap(r, hi), column = narrow column i)
  float* dw = P.dw_part + (size_t)blockIdx.x * wn * nn;
  if (P.wide_is_out) {                               // dW[out = wide][in = narrow], db[out]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (nin) dw[(size_t)o * nn + i] = acc[r];
      else if (ones) P.db_part[(size_t)blockIdx.x * wn + o] = acc[r];
    }
  } else {                                           // dW[out = narrow][in = wide], db[out] = sum of the narrow operand
    if (nin) {
#pragma unroll
      for (int r = 0; r < 16; ++r) dw[(size_t)i * wn + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * hi] = acc[r];
    }
    if (P.db_part && wave == 0) {
      nsum += __shfl_xor(nsum, 32, 64);              // even + odd rows
      if (nin && hi == 0) P.db_part[(size_t)blockIdx.x * nn + i] = nsum;
    }
  }
}

template <int GATE, int KPX>
static int launch_bwd_weight_multi_kp(GemmDev& g, int max_tiles, int max_splits, hipStream_t s) {
  constexpr int KP = KPX ? KPX : gemm_panel<1>();
  constexpr int lds = 2 * (int)sizeof(float) * (tile_floats<false, 64, KP>() + tile_floats<false, 64, KP>());
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_f32_kernel<true, false, GATE, 0, 2, 1, KPX>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { trl_set_error("gemm: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_f32_kernel<true, false, GATE, 0, 2, 1, KPX>), dim3(max_tiles, g.groups, max_splits), dim3(256), lds, s, g);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}
template <int GATE>
static int launch_bwd_weight_multi(GemmDev& g, int max_tiles, int max_splits, hipStream_t s) {
  // slices of <= 512 batch rows on >= 768 workgroups: the 32-wide panel, four workgroups per CU (see gemm_f32_kernel)
  static const int pin = [] { const char* e = getenv("TRL_GEMM_KP"); return e ? atoi(e) : 0; }();
  int64_t wgs = 0;
  int longest = 0;
  for (int i = 0; i < g.groups; ++i) { wgs += (int64_t)g.shp[i].tiles * g.shp[i].splits; longest = std::max(longest, g.shp[i].split_len); }
  if (pin != 64 && (pin == 32 || (wgs >= 768 && longest <= 512)))
    return launch_bwd_weight_multi_kp<GATE, 32>(g, max_tiles, max_splits, s);
  return launch_bwd_weight_multi_kp<GATE, 0>(g, max_tiles, max_splits, s);
}
extern "C" int trl_linear_bwd_weight_partials_multi_f32(int G, const float* const* dy, const float* const* y_gate,
                                                        int gate_act, const float* const* x, const int* K, const int* N,
                                                        int want_db, float* const* workspace, int M, void* stream) {
  TRL_REQUIRE(G >= 1 && G <= GEMM_MAX_GROUPS, "1..12 problems per launch");
  TRL_REQUIRE(dy && x && K && N && workspace && M > 0, "null pointer array / empty batch");
  TRL_REQUIRE(gate_act == TRL_ACT_TANH || gate_act == TRL_ACT_RELU || gate_act == TRL_ACT_NONE, "unknown activation");
  bool all_skinny = true;
  for (int i = 0; i < G; ++i) all_skinny = all_skinny && K[i] > 0 && N[i] > 0 && skinny_layer(K[i], N[i]);
  if (all_skinny) {
    SkinnyDev sk{};
    sk.M = M; sk.gate_act = gate_act;
    int max_splits = 0, max_wide = 0;
    for (int i = 0; i < G; ++i) {
      TRL_REQUIRE(dy[i] && x[i] && workspace[i], "null pointer / bad layer size");
      const int split_len = multi_split_len(M, K[i], N[i]), splits = trl_ceil_div(M, split_len);
      const bool wide_out = K[i] <= 31;              // few inputs: dZ (M x N) is the wide operand
      float* part = workspace[i];
      const float* gate = (y_gate && gate_act != TRL_ACT_NONE) ? y_gate[i] : nullptr;
      sk.p[i] = SkinnyProb{wide_out ? dy[i] : x[i], wide_out ? x[i] : dy[i], gate, part,
                           want_db ? part + (size_t)splits * N[i] * K[i] : nullptr,
                           wide_out ? N[i] : K[i], wide_out ? K[i] : N[i], wide_out ? 1 : 0, split_len, splits};
      max_splits = std::max(max_splits, splits); max_wide = std::max(max_wide, wide_out ? N[i] : K[i]);
    }
    bool any = false;
    for (int i = 0; i < G; ++i) any = any || sk.p[i].gate != nullptr;
    const dim3 grid(max_splits, G), block(2 * max_wide);
    if (any && gate_act == TRL_ACT_TANH)      hipLaunchKernelGGL(skinny_bwdw_kernel<TRL_ACT_TANH>, grid, block, 0, (hipStream_t)stream, sk);
    else if (any && gate_act == TRL_ACT_RELU) hipLaunchKernelGGL(skinny_bwdw_kernel<TRL_ACT_RELU>, grid, block, 0, (hipStream_t)stream, sk);
    else                                      hipLaunchKernelGGL(skinny_bwdw_kernel<TRL_ACT_NONE>, grid, block, 0, (hipStream_t)stream, sk);
    TRL_LAUNCH_CHECK();
    return TRL_OK;
  }
  GemmDev g{};
  g.gate_act = gate_act; g.act = TRL_ACT_NONE; g.groups = G; g.hetero = 1;
  int max_tiles = 0, max_splits = 0;
  bool any_gate = false;
  for (int i = 0; i < G; ++i) {
    TRL_REQUIRE(dy[i] && x[i] && workspace[i] && K[i] > 0 && N[i] > 0, "null pointer / bad layer size");
    const int split_len = multi_split_len(M, K[i], N[i]);
    const int splits = trl_ceil_div(M, split_len);
    GemmShape& h = g.shp[i];
    h.M = N[i]; h.N = K[i]; h.K = M; h.lda = N[i]; h.ldb = K[i]; h.ldc = K[i];
    h.split_len = splits > 1 ? split_len : M; h.splits = splits;
    h.tiles_n = trl_ceil_div(K[i], 64); h.tiles = h.tiles_n * trl_ceil_div(N[i], 64);
    float* part = workspace[i];
    float* cpart = want_db ? part + (size_t)splits * N[i] * K[i] : nullptr;
    const float* gate = (y_gate && gate_act != TRL_ACT_NONE) ? y_gate[i] : nullptr;
    any_gate |= gate != nullptr;
    g.grp[i] = GemmGroup{dy[i], x[i], part, nullptr, gate, cpart};
    max_tiles = std::max(max_tiles, h.tiles); max_splits = std::max(max_splits, splits);
  }
  // one z-slice in the grid would read "no split" in the kernel: the shapes' own split_len already cover that case
  g.A = g.grp[0].A; g.B = g.grp[0].B; g.C = g.grp[0].C; g.a_gate = g.grp[0].a_gate; g.colsum = g.grp[0].colsum;
  g.M = g.shp[0].M; g.N = g.shp[0].N; g.K = M; g.lda = g.shp[0].lda; g.ldb = g.shp[0].ldb; g.ldc = g.shp[0].ldc;
  g.split_len = g.shp[0].split_len; g.tiles_n = g.shp[0].tiles_n; g.tiles = g.shp[0].tiles;
  hipStream_t s = (hipStream_t)stream;
  if (!any_gate) return launch_bwd_weight_multi<TRL_ACT_NONE>(g, max_tiles, max_splits, s);
  if (gate_act == TRL_ACT_TANH) return launch_bwd_weight_multi<TRL_ACT_TANH>(g, max_tiles, max_splits, s);
  return launch_bwd_weight_multi<TRL_ACT_RELU>(g, max_tiles, max_splits, s);
}
// ---- first conv layer on uint8 frames as an implicit GEMM ----
static int fill_conv(const char* who, const uint8_t* frames, int B, int C, int H, int W, int kh, int kw, int sh, int sw,
                     float scale, float shift, ConvSrc& cv, int& M, int& K) {
  if (!(B > 0 && C > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0 && H >= kh && W >= kw)) {
    trl_set_error("%s: bad geometry", who); return TRL_EINVAL;
  }
  if ((kw & 3) || (sw & 3) || (W & 3) || (reinterpret_cast<uintptr_t>(frames) & 3)) {
    trl_set_error("%s: the implicit-GEMM path needs kw, sw and W to be multiples of 4 and 4-byte aligned frames "
                  "(use trl_im2col_u8_nchw + trl_linear_* otherwise)", who);
    return TRL_EINVAL;
  }
  if ((int64_t)B * C * H * W >= ((int64_t)1 << 32) - 1) { trl_set_error("%s: frame batch exceeds 4 GiB", who); return TRL_EINVAL; }
  cv.frames = frames; cv.x = nullptr; cv.C = C; cv.H = H; cv.W = W; cv.kh = kh; cv.kw = kw; cv.sh = sh; cv.sw = sw;
  cv.Ho = (H - kh) / sh + 1; cv.Wo = (W - kw) / sw + 1; cv.scale = scale; cv.shift = shift;
  fastdiv_gen((uint32_t)(cv.Ho * cv.Wo), cv.hw_magic, cv.hw_shift);
  fastdiv_gen((uint32_t)cv.Wo, cv.w_magic, cv.w_shift);
  const int64_t m = (int64_t)B * cv.Ho * cv.Wo;
  if (m >= ((int64_t)1 << 31)) { trl_set_error("%s: too many output positions", who); return TRL_EINVAL; }
  M = (int)m; K = C * kh * kw;
  return TRL_OK;
}

__global__ __launch_bounds__(256) void conv_perm_kernel(PermJobs pj) {
  conv_perm_jobs(pj, blockIdx.x, gridDim.x, threadIdx.x, 256);
}
static int conv_fwd_u8(const uint8_t* frames, const float* w, const float* bias, float* y, const uint8_t* frames2,
                       const float* w2, const float* bias2, float* y2, int B, int C, int H, int W, int kh, int kw, int sh,
                       int sw, float scale, float shift, int Cout, int act, const trl_conv_riders_t* riders, void* stream);
extern "C" int trl_conv_fwd_u8_f32(const uint8_t* frames, const float* w, const float* bias, float* y, int B, int C, int H,
                                   int W, int kh, int kw, int sh, int sw, float scale, float shift, int Cout, int act,
                                   const trl_conv_riders_t* riders, void* stream) {
  return conv_fwd_u8(frames, w, bias, y, nullptr, nullptr, nullptr, nullptr, B, C, H, W, kh, kw, sh, sw, scale, shift, Cout,
                     act, riders, stream);
}
// The first conv layer of TWO networks of one architecture on two frame batches of one shape -- DQN's online net on obs and
// target net on next_obs (torchrl/algo/off_policy/dqn.py:38-52) -- as ONE launch; the riders may carry both networks' jobs.
extern "C" int trl_conv_fwd_u8_pair_f32(const uint8_t* frames_a, const float* w_a, const float* bias_a, float* y_a,
                                        const uint8_t* frames_b, const float* w_b, const float* bias_b, float* y_b, int B,
                                        int C, int H, int W, int kh, int kw, int sh, int sw, float scale, float shift,
                                        int Cout, int act, const trl_conv_riders_t* riders, void* stream) {
  TRL_REQUIRE(frames_b && w_b && y_b, "conv_fwd_u8_pair: null pointer of the second problem");
  TRL_REQUIRE((bias_a == nullptr) == (bias_b == nullptr), "conv_fwd_u8_pair: both problems with or both without a bias");
  return conv_fwd_u8(frames_a, w_a, bias_a, y_a, frames_b, w_b, bias_b, y_b, B, C, H, W, kh, kw, sh, sw, scale, shift, Cout,
                     act, riders, stream);
}
static int conv_fwd_u8(const uint8_t* frames, const float* w, const float* bias, float* y, const uint8_t* frames2,
                       const float* w2, const float* bias2, float* y2, int B, int C, int H, int W, int kh, int kw, int sh,
                       int sw, float scale, float shift, int Cout, int act, const trl_conv_riders_t* riders, void* stream) {
  TRL_REQUIRE(frames && w && y && Cout > 0, "null pointer / bad Cout");
  TRL_REQUIRE(act == TRL_ACT_TANH || act == TRL_ACT_RELU || act == TRL_ACT_NONE, "unknown activation");
  PermJobs pj{};
  if (riders) {
    TRL_REQUIRE(riders->n_perm >= 0 && riders->n_perm <= CONV_PERM_MAX && riders->n_dx >= 0 && riders->n_dx <= CONV_PERM_MAX,
                "conv_fwd_u8: 0..4 jobs of either kind");
    pj.n = riders->n_perm; pj.n_dx = riders->n_dx;
    for (int k = 0; k < pj.n; ++k) {
      TRL_REQUIRE(riders->perm_src[k] && riders->perm_dst[k] && riders->perm_cout[k] > 0 && riders->perm_c[k] > 0 && riders->perm_khw[k] > 0,
                  "conv_fwd_u8: bad re-ordering job");
      pj.src[k] = riders->perm_src[k]; pj.dst[k] = riders->perm_dst[k]; pj.cout[k] = riders->perm_cout[k];
      pj.c[k] = riders->perm_c[k]; pj.khw[k] = riders->perm_khw[k];
    }
    for (int k = 0; k < pj.n_dx; ++k) {
      TRL_REQUIRE(riders->dx_w[k] && riders->dx_ws[k] &&
                  trl_conv_bwd_input_nhwc_ok(riders->dx_cin[k], riders->dx_cout[k], riders->dx_kh[k], riders->dx_kw[k], riders->dx_sh[k], riders->dx_sw[k]),
                  "conv_fwd_u8: dx job outside trl_conv_bwd_input_nhwc_ok");
      pj.dx_w[k] = riders->dx_w[k]; pj.dx_ws[k] = riders->dx_ws[k];
      pj.dx_g[k] = DxGeom{0, riders->dx_cin[k], 0, 0, riders->dx_kh[k], riders->dx_kw[k], riders->dx_sh[k], riders->dx_sw[k], 0, 0,
                          riders->dx_cout[k], TRL_ACT_NONE, TRL_ACT_NONE, 1, 0};
    }
  }
  GemmDev g{};
  int M, K;
  int rc = fill_conv("conv_fwd_u8", frames, B, C, H, W, kh, kw, sh, sw, scale, shift, g.cv, M, K);
  if (rc) return rc;
  if (trl_conv1_direct_ok(K, Cout, w) && (!frames2 || trl_conv1_direct_ok(K, Cout, w2)))   // narrow first layer: register-weights
    return trl_conv1_direct_fwd(g.cv, w, bias, y, M, K, Cout, act, pj, (hipStream_t)stream, frames2, w2, bias2, y2);   // kernel, no LDS staging
  if (pj.n || pj.n_dx) {                             // (the generic kernel carries no riders: a launch of their own)
    hipLaunchKernelGGL(conv_perm_kernel, dim3(CONV_PERM_BLOCKS), dim3(256), 0, (hipStream_t)stream, pj);
    TRL_LAUNCH_CHECK();
  }
  g.A = nullptr; g.B = w; g.C = y; g.bias = bias; g.a_gate = nullptr; g.M = M; g.N = Cout; g.K = K;
  g.lda = K; g.ldb = K; g.ldc = Cout; g.act = act; g.gate_act = TRL_ACT_NONE; g.split_len = K; g.colsum = nullptr;
  rc = launch_gemm<false, true, 1>(g, 1, (hipStream_t)stream);
  if (rc || !frames2) return rc;
  g.cv.frames = frames2; g.B = w2; g.C = y2; g.bias = bias2;       // (wide first layers: the second problem as a launch of its own)
  return launch_gemm<false, true, 1>(g, 1, (hipStream_t)stream);
}

static int fill_conv_nhwc(const char* who, const float* x, int B, int C, int H, int W, int kh, int kw, int sh, int sw,
                          ConvSrc& cv, int& M, int& K) {
  if (!(B > 0 && C > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0 && H >= kh && W >= kw)) {
    trl_set_error("%s: bad geometry", who); return TRL_EINVAL;
  }
  if ((C & 3) || (reinterpret_cast<uintptr_t>(x) & 15)) {
    trl_set_error("%s: the implicit-GEMM path needs C %% 4 == 0 and 16-byte aligned activations "
                  "(use trl_im2col_f32 + trl_linear_* otherwise)", who);
    return TRL_EINVAL;
  }
  if ((int64_t)B * C * H * W >= ((int64_t)1 << 32) - 1) { trl_set_error("%s: activation tensor too large", who); return TRL_EINVAL; }
  cv.x = x; cv.frames = nullptr; cv.C = C; cv.H = H; cv.W = W; cv.kh = kh; cv.kw = kw; cv.sh = sh; cv.sw = sw;
  cv.Ho = (H - kh) / sh + 1; cv.Wo = (W - kw) / sw + 1; cv.scale = 1.0f; cv.shift = 0.0f;
  fastdiv_gen((uint32_t)(cv.Ho * cv.Wo), cv.hw_magic, cv.hw_shift);
  fastdiv_gen((uint32_t)cv.Wo, cv.w_magic, cv.w_shift);
  const int64_t m = (int64_t)B * cv.Ho * cv.Wo;
  if (m >= ((int64_t)1 << 31)) { trl_set_error("%s: too many output positions", who); return TRL_EINVAL; }
  M = (int)m; K = C * kh * kw;
  return TRL_OK;
}

extern "C" int trl_conv_fwd_nhwc_f32(const float* x, const float* w, const float* bias, float* y, int B, int C, int H, int W,
                                     int kh, int kw, int sh, int sw, int Cout, int act, int out_chw, int w_perm, void* stream) {
  TRL_REQUIRE(x && w && y && Cout > 0, "null pointer / bad Cout");
  TRL_REQUIRE(act == TRL_ACT_TANH || act == TRL_ACT_RELU || act == TRL_ACT_NONE, "unknown activation");
  GemmDev g{};
  int M, K;
  int rc = fill_conv_nhwc("conv_fwd_nhwc", x, B, C, H, W, kh, kw, sh, sw, g.cv, M, K);
  if (rc) return rc;
  g.A = nullptr; g.B = w; g.C = y; g.bias = bias; g.a_gate = nullptr; g.M = M; g.N = Cout; g.K = K;
  g.lda = K; g.ldb = K; g.ldc = Cout; g.act = act; g.gate_act = TRL_ACT_NONE; g.split_len = K; g.colsum = nullptr;
  g.chw_p = out_chw ? g.cv.Ho * g.cv.Wo : 0;
  g.b_perm = w_perm ? 1 : 0;
  return launch_gemm<false, true, 3>(g, 1, (hipStream_t)stream);
}

// G same-geometry conv layers (different inputs, weights, outputs) in one launch: the online and the target network of
// a DQN update run the same trunk on obs and next_obs
extern "C" int trl_conv_fwd_nhwc_group_f32(int G, const float* const* x, const float* const* w, const float* const* bias,
                                           float* const* y, int B, int C, int H, int W, int kh, int kw, int sh, int sw,
                                           int Cout, int act, int out_chw, int w_perm, void* stream) {
  TRL_REQUIRE(G >= 1 && G <= GEMM_MAX_GROUPS, "1..12 problems per grouped launch");
  TRL_REQUIRE(x && w && y && Cout > 0, "null pointer array / bad Cout");
  TRL_REQUIRE(act == TRL_ACT_TANH || act == TRL_ACT_RELU || act == TRL_ACT_NONE, "unknown activation");
  GemmDev g{};
  int M, K;
  for (int i = 0; i < G; ++i) {
    TRL_REQUIRE(x[i] && w[i] && y[i], "null pointer");
    int rc = fill_conv_nhwc("conv_fwd_nhwc_group", x[i], B, C, H, W, kh, kw, sh, sw, g.cv, M, K);   // checks every input
    if (rc) return rc;
    g.grp[i] = GemmGroup{x[i], w[i], y[i], bias ? bias[i] : nullptr, nullptr, nullptr};
  }
  g.groups = G;
  g.A = nullptr; g.B = w[0]; g.C = y[0]; g.bias = bias ? bias[0] : nullptr; g.a_gate = nullptr; g.M = M; g.N = Cout; g.K = K;
  g.lda = K; g.ldb = K; g.ldc = Cout; g.act = act; g.gate_act = TRL_ACT_NONE; g.split_len = K; g.colsum = nullptr;
  g.cv.x = x[0];
  g.chw_p = out_chw ? g.cv.Ho * g.cv.Wo : 0;
  g.b_perm = w_perm ? 1 : 0;
  return launch_gemm<false, true, 3>(g, 1, (hipStream_t)stream);
}

extern "C" int trl_conv_bwd_weight_nhwc_f32(const float* dy, const float* y_gate, int gate_act, const float* x, float* dw,
                                            float* db, float* workspace, int B, int C, int H, int W, int kh, int kw, int sh,
                                            int sw, int Cout, void* stream) {
  TRL_REQUIRE(dy && x && dw && workspace && Cout > 0, "null pointer / bad Cout");
  ConvSrc cv{};
  int M, K;
  int rc = fill_conv_nhwc("conv_bwd_weight_nhwc", x, B, C, H, W, kh, kw, sh, sw, cv, M, K);
  if (rc) return rc;
  return bwd_weight_impl<4>(1, &dy, &y_gate, gate_act, nullptr, &cv, &dw, &db, workspace, M, K, Cout, (hipStream_t)stream);
}

extern "C" int trl_conv_bwd_weight_workspace(int B, int C, int H, int W, int kh, int kw, int sh, int sw, int Cout) {
  if (!(B > 0 && C > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0 && H >= kh && W >= kw && Cout > 0)) return TRL_EINVAL;
  const int64_t m = (int64_t)B * ((H - kh) / sh + 1) * ((W - kw) / sw + 1);
  if (m >= ((int64_t)1 << 31)) return TRL_EINVAL;
  const int K = C * kh * kw;
  const int generic = trl_linear_bwd_weight_workspace((int)m, K, Cout);
  // the uint8 entry point may take the direct kernel (weights are not known here: size for both)
  const int direct = (Cout <= 16 && K <= 256 && (K & 63) == 0) ? trl_conv1_direct_bwdw_workspace((int)m, K, Cout) : 0;
  return std::max(generic, direct);
}

extern "C" int trl_conv_bwd_weight_u8_f32(const float* dy, const float* y_gate, int gate_act, const uint8_t* frames,
                                          float* dw, float* db, float* workspace, int B, int C, int H, int W, int kh,
                                          int kw, int sh, int sw, float scale, float shift, int Cout, void* stream) {
  TRL_REQUIRE(dy && frames && dw && workspace && Cout > 0, "null pointer / bad Cout");
  ConvSrc cv{};
  int M, K;
  int rc = fill_conv("conv_bwd_weight_u8", frames, B, C, H, W, kh, kw, sh, sw, scale, shift, cv, M, K);
  if (rc) return rc;
  if (trl_conv1_direct_ok(K, Cout, nullptr))           // (the 16-byte alignment is a requirement of the forward's weight loads)
    return trl_conv1_direct_bwdw(cv, dy, y_gate, gate_act, dw, db, workspace, M, K, Cout, (hipStream_t)stream);
  return bwd_weight_impl<2>(1, &dy, &y_gate, gate_act, nullptr, &cv, &dw, &db, workspace, M, K, Cout, (hipStream_t)stream);
}
