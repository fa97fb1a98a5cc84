// K10c -- input gradient of a conv layer on channels-last activations as an IMPLICIT transposed convolution.
//
// Replaces, for the conv trunk of CNNBase (torchrl/networks/base.py:59-107; autograd's conv2d backward w.r.t. the
// input in the reference), the pair  dcols = dZ . W  (a (B Ho Wo) x (Cin kh kw) matrix: 42 MB at cfg 5's second layer,
// written and read back) + gather-form col2im.  Here the cols matrix never exists:
//     dX[b, y, x, c] = sum over taps (i, j) with i = y (mod sh), j = x (mod sw), and over co, of
//                      dZ[b, (y - i) / sh, (x - j) / sw, co] * W[co, c, i, j],        dZ = dY * act'(Y).
// Pixels with the same (y mod sh, x mod sw) -- a parity CLASS -- see the same taps, so each class is a dense product
// (positions of the class) x (taps_of_class * Cout) . (taps_of_class * Cout) x Cin.
//   * a prep launch re-orders the weights once per call into [class][tap][MFMA B-operand order] (a few thousand
//     floats), so that a workgroup stages its class's block into LDS with contiguous 16-byte loads and reads it
//     lane-linearly (conflict free);
//   * a workgroup = one class (blockIdx.y) x a run of 64-position tiles; wave w owns positions [16 w, 16 w + 16) of a
//     tile and all Cin / 16 column blocks (v_mfma_f32_16x16x4_f32: rows = positions, columns = input channels,
//     k = 4 output channels);
//   * the A operand (gated dZ) comes straight from global / L1 with 16-byte loads, 4 - 8 taps x all Cout in flight at
//     once.  A tile's footprint of dZ is ~10 KB, re-read once per tap: L1 hits.
#include <algorithm>
#include <cstdlib>
#include "trl_common.h"
#include "trl_mlp.h"

#include "trl_conv.h"

__device__ __forceinline__ float dx_dact(int act, float y) {
  if (act == TRL_ACT_TANH) return 1.0f - y * y;
  if (act == TRL_ACT_RELU) return y > 0.0f ? 1.0f : 0.0f;
  return 1.0f;
}
// n / d for 0 <= n < 2^23 through the float reciprocal (runtime integer division is ~45 VALU instructions, and a lane
// decodes five positions per tile)
__device__ __forceinline__ int dx_div(int n, int d, float inv) {
  int q = (int)((float)n * inv);
  int r = n - q * d;
  q += (r >= d) - (r < 0);
  return q;
}
// wprep[class block][tap][co / 16][r][cb][gq][j] = W[co = 16 chunk + 4 gq + r][c = 16 cb + j][i][j_tap]: MFMA step
// (chunk, r) of column block cb reads 64 consecutive floats.  Class blocks follow each other in class order.
__global__ __launch_bounds__(256) void conv_dx_prep_kernel(const float* __restrict__ w, float* __restrict__ wprep, DxGeom g) {
  dx_prep_range(w, wprep, g, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

// the re-ordered weights of SEVERAL layers in one launch (blockIdx.y = layer): a backward pass through a conv trunk needs
// every layer's, and each prep launch of its own is ~5 us of dependent launch for a few KB
#define DX_PREP_MAX 8
struct DxPrepSet { const float* w[DX_PREP_MAX]; float* wprep[DX_PREP_MAX]; DxGeom g[DX_PREP_MAX]; };
__global__ __launch_bounds__(256) void conv_dx_prep_multi_kernel(DxPrepSet p) {
  const DxGeom g = p.g[blockIdx.y];
  dx_prep_range(p.w[blockIdx.y], p.wprep[blockIdx.y], g, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

template <int CB, int NCH, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void conv_dx_kernel(const float* __restrict__ dy, const float* __restrict__ yg,
                                                      const float* __restrict__ wprep, float* __restrict__ dx,
                                                      const float* __restrict__ xg, DxGeom g) {
  extern __shared__ __attribute__((aligned(16))) float Ws[];
  constexpr int TAP = 16 * NCH * 16 * CB;                        // floats of one tap's weights
  const int py = blockIdx.y / g.sw, px = blockIdx.y - py * g.sw;
  const int Hc = py < g.H ? (g.H - py + g.sh - 1) / g.sh : 0, Wc = px < g.W ? (g.W - px + g.sw - 1) / g.sw : 0;
  const int rows = g.B * Hc * Wc;
  constexpr int TILE = 16 * WAVES, THREADS = 64 * WAVES;
  int row0 = blockIdx.x * g.tiles_per_wg * TILE;
  if (row0 >= rows) return;
  const int nti = py < g.kh ? (g.kh - py + g.sh - 1) / g.sh : 0, ntj = px < g.kw ? (g.kw - px + g.sw - 1) / g.sw : 0;
  const int ntap = nti * ntj;
  {
    int off = 0;
    for (int cls = 0; cls < (int)blockIdx.y; ++cls) off += dx_class_taps(g, cls / g.sw, cls % g.sw) * TAP;
    const f32x4* src = reinterpret_cast<const f32x4*>(wprep + off);
    f32x4* dst = reinterpret_cast<f32x4*>(Ws);
#ifdef TRL_EXP_DX
    const int n4 = (g.dbg & 1) ? 0 : ntap * (TAP / 4);
#else
    const int n4 = ntap * (TAP / 4);
#endif
    for (int e = threadIdx.x; e < n4; e += 8 * THREADS) {        // 8 loads in flight per thread
      f32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) if (e + THREADS * k < n4) v[k] = src[e + THREADS * k];
#pragma unroll
      for (int k = 0; k < 8; ++k) if (e + THREADS * k < n4) dst[e + THREADS * k] = v[k];
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, gq = lane >> 4;
  const bool gated = yg != nullptr && g.gate_act != TRL_ACT_NONE;
  const float inv_wc = 1.0f / (float)Wc, inv_hc = 1.0f / (float)Hc, inv_ntj = 1.0f / (float)ntj;
  for (int tile = 0; tile < g.tiles_per_wg && row0 < rows; ++tile, row0 += TILE) {
    const int row = row0 + 16 * wave + j;                        // A row of this lane
    const bool row_ok = row < rows;
    int b = 0, yq = 0, xq = 0;
    if (row_ok) { const int t = dx_div(row, Wc, inv_wc); xq = row - t * Wc; b = dx_div(t, Hc, inv_hc); yq = t - b * Hc; }
    f32x4 acc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc[cb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // Taps in groups of G: every load of a group (G taps x all Cout of dY and of the gate) is issued before the
    // group's MFMAs -- one memory round trip per group instead of one per tap (a tap's 32 MFMAs are 0.4 us, a
    // round trip under load ~2 us).
    constexpr int G = 8 / NCH;
    f32x4 a[G][NCH], y[G][NCH];
    for (int t0 = 0; t0 < ntap; t0 += G) {
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int t = t0 + u;
        if (t >= ntap) break;
        const int ti = dx_div(t, ntj, inv_ntj), tj = t - ti * ntj;
        const int oy = yq - ti, ox = xq - tj;
#ifdef TRL_EXP_DX
        const bool ok = !(g.dbg & 2) && row_ok && oy >= 0 && oy < g.Ho && ox >= 0 && ox < g.Wo;
#else
        const bool ok = row_ok && oy >= 0 && oy < g.Ho && ox >= 0 && ox < g.Wo;
#endif
        const size_t base = ok ? (((size_t)b * g.Ho + oy) * g.Wo + ox) * g.Cout + 4 * gq : 0;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
          a[u][ch] = ok ? *reinterpret_cast<const f32x4*>(dy + base + 16 * ch) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
          if (gated) y[u][ch] = ok ? *reinterpret_cast<const f32x4*>(yg + base + 16 * ch) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
      }
#pragma unroll
      for (int u = 0; u < G; ++u) {
        if (t0 + u >= ntap) break;
#ifdef TRL_EXP_DX
        if (g.dbg & 8) { acc[0][0] += a[u][0][0]; continue; }
#endif
        const float* wt = Ws + (t0 + u) * TAP + lane;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
          f32x4 av = a[u][ch];
          if (gated) {
#pragma unroll
            for (int r = 0; r < 4; ++r) av[r] *= dx_dact(g.gate_act, y[u][ch][r]);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
              acc[cb] = mfma16(av[r], wt[((ch * 4 + r) * CB + cb) * 64], acc[cb]);
        }
      }
    }
    // C reg r of lane (j, gq): position 4 gq + r of the wave's block, input channel 16 cb + j
    const int orow0 = row0 + 16 * wave + 4 * gq;
    const int t0o = dx_div(orow0, Wc, inv_wc);
    int xo = orow0 - t0o * Wc, bo = dx_div(t0o, Hc, inv_hc), yo = t0o - bo * Hc;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (orow0 + r >= rows) break;
#ifdef TRL_EXP_DX
      if ((g.dbg & 4) && acc[0][r] != 12345.0f) continue;
#endif
      if (r > 0 && ++xo == Wc) { xo = 0; if (++yo == Hc) { yo = 0; ++bo; } }   // the next position of the class
      const size_t o = (((size_t)bo * g.H + (g.sh * yo + py)) * g.W + (g.sw * xo + px)) * g.Cin + j;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        float v = acc[cb][r];
        if (xg) v *= dx_dact(g.x_gate_act, xg[o + 16 * cb]);     // hand the previous layer its dZ, not its dY
        dx[o + 16 * cb] = v;
      }
    }
  }
}


// ---- image-tile form: the gated dZ of a few whole images is staged in LDS ONCE and every tap reads it from there ----
// The class form above re-reads a tile's dZ (and its gate) through L1 once per tap: at cfg 5 that is ~64 B/clk/CU of
// half-used cache lines -- the L1's whole bandwidth -- for 32 MFMAs per tap, and the activation derivative is recomputed
// per tap (rocprofv3 round 4: 0.24 / 0.33 MFMA-busy, SQ_WAIT_INST_ANY 1.3x SQ_ACTIVE_INST_ANY).  Here a workgroup owns
// `img` whole images: it stages ALL classes' re-ordered weights and dZ = dY * act'(Y) of its images (rows padded to
// Cout + 4 floats: the 16 lanes of a ds_read_b128 group hit 16 different bank quads), then walks the parity classes;
// a wave takes 16-position blocks of a class, its A operand is one 16-byte LDS read per (tap, 16 output channels).
// Same MFMA sequence per output element as the class form (taps ascending, then channels): bit-identical results.
struct DxPad { int pt, pl, Hp, Wp; };                  // zero border of an image's dZ in LDS: rows above / columns left, padded size
__host__ __device__ inline DxPad dx_pad(const DxGeom& g) {
  // tap (ti, tj) of the position (yq, xq) of a class reads dZ[yq - ti][xq - tj]: yq - ti runs from -(ceil(kh / sh) - 1) to
  // ceil(H / sh) - 1, of which [0, Ho) exists -- the rest is the border
  const int pt = (g.kh + g.sh - 1) / g.sh - 1, pl = (g.kw + g.sw - 1) / g.sw - 1;
  const int pb = std::max(0, (g.H + g.sh - 1) / g.sh - g.Ho), pr = std::max(0, (g.W + g.sw - 1) / g.sw - g.Wo);
  return DxPad{pt, pl, g.Ho + pt + pb, g.Wo + pl + pr};
}

template <int CB, int NCH, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void conv_dx_img_kernel(const float* __restrict__ dy, const float* __restrict__ yg,
                                                          const float* __restrict__ wprep, float* __restrict__ dx,
                                                          const float* __restrict__ xg, DxGeom g, int img_per_wg) {
  extern __shared__ __attribute__((aligned(16))) float Ws[];
  constexpr int COUT = 16 * NCH, LDZ = COUT + 4, TAP = 16 * NCH * 16 * CB, THREADS = 64 * WAVES, Q = COUT / 4;
  const int wfloats = COUT * 16 * CB * g.kh * g.kw;
  float* Zs = Ws + wfloats;
  const int img0 = blockIdx.x * img_per_wg;
  const int nimg = min(img_per_wg, g.B - img0);
  const DxPad pd = dx_pad(g);
  const bool gated = yg != nullptr && g.gate_act != TRL_ACT_NONE;
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(wprep);
    f32x4* dst = reinterpret_cast<f32x4*>(Ws);
    const int n4 = wfloats / 4;
    for (int e = threadIdx.x; e < n4; e += 8 * THREADS) {        // 8 loads in flight per thread
      f32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) if (e + THREADS * k < n4) v[k] = src[e + THREADS * k];
#pragma unroll
      for (int k = 0; k < 8; ++k) if (e + THREADS * k < n4) dst[e + THREADS * k] = v[k];
    }
    // the images' dZ = dY * act'(Y) with a ZERO BORDER (pd): a tap that falls outside the layer's output reads zeros, so
    // the tap walk needs no range test and no select -- one scalar offset per tap on top of a per-block lane address
    const int cells = nimg * pd.Hp * pd.Wp, nz = cells * Q;      // 16-byte pieces
    const float inv_q = 1.0f / (float)Q, inv_wp = 1.0f / (float)pd.Wp, inv_hp = 1.0f / (float)pd.Hp;
    const f32x4* d4 = reinterpret_cast<const f32x4*>(dy) + (size_t)img0 * g.Ho * g.Wo * Q;
    const f32x4* y4 = reinterpret_cast<const f32x4*>(yg) + (size_t)img0 * g.Ho * g.Wo * Q;
    for (int e = threadIdx.x; e < nz; e += 4 * THREADS) {
      f32x4 v[4], y[4];
      int cell[4], q[4];
      bool in[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int ee = e + THREADS * k;
        in[k] = false;
        if (ee < nz) {
          cell[k] = dx_div(ee, Q, inv_q); q[k] = ee - cell[k] * Q;
          const int t = dx_div(cell[k], pd.Wp, inv_wp), xp = cell[k] - t * pd.Wp, b = dx_div(t, pd.Hp, inv_hp), yp = t - b * pd.Hp;
          const int oy = yp - pd.pt, ox = xp - pd.pl;
          in[k] = oy >= 0 && oy < g.Ho && ox >= 0 && ox < g.Wo;
          const int srcp = in[k] ? ((b * g.Ho + oy) * g.Wo + ox) * Q + q[k] : 0;
          v[k] = d4[srcp];
          if (gated) y[k] = y4[srcp];
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int ee = e + THREADS * k;
        if (ee >= nz) break;
        if (gated) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[k][r] *= dx_dact(g.gate_act, y[k][r]);
        }
        *reinterpret_cast<f32x4*>(Zs + cell[k] * LDZ + 4 * q[k]) = in[k] ? v[k] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      }
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, gq = lane >> 4;
  int woff = 0;
  for (int cls = 0; cls < g.sh * g.sw; ++cls) {
    const int py = cls / g.sw, px = cls - py * g.sw;
    const int Hc = py < g.H ? (g.H - py + g.sh - 1) / g.sh : 0, Wc = px < g.W ? (g.W - px + g.sw - 1) / g.sw : 0;
    const int nti = py < g.kh ? (g.kh - py + g.sh - 1) / g.sh : 0, ntj = px < g.kw ? (g.kw - px + g.sw - 1) / g.sw : 0;
    const int ntap = nti * ntj, rows = nimg * Hc * Wc;
    const float* Wc_lds = Ws + woff;
    woff += ntap * TAP;
    if (rows <= 0 || ntap <= 0) continue;
    const float inv_wc = 1.0f / (float)Wc, inv_hc = 1.0f / (float)Hc;
    for (int row0 = 16 * wave; row0 < rows; row0 += 16 * WAVES) {
      // A row of this lane; rows past the end read row 0's cells: an MFMA output row depends on its own A row only, and
      // those output rows are never stored
      const int row = row0 + j < rows ? row0 + j : 0;
      const int t = dx_div(row, Wc, inv_wc), xq = row - t * Wc, b = dx_div(t, Hc, inv_hc), yq = t - b * Hc;
      const float* zlane = Zs + (((b * pd.Hp + yq + pd.pt) * pd.Wp) + xq + pd.pl) * LDZ + 4 * gq;
      // C reg r of lane (j, gq): position 4 gq + r of the wave's block, input channel 16 cb + j.  The output addresses are
      // decoded and the gate of the layer below (x_gate) is requested BEFORE the tap walk: its global round trip runs
      // under the MFMAs instead of behind them
      const int orow0 = row0 + 4 * gq;
      size_t oaddr[4];
      float xgv[4][CB];
      {
        const int t0o = dx_div(orow0 < rows ? orow0 : 0, Wc, inv_wc);
        int xo = (orow0 < rows ? orow0 : 0) - t0o * Wc, bo = dx_div(t0o, Hc, inv_hc), yo = t0o - bo * Hc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (r > 0 && ++xo == Wc) { xo = 0; if (++yo == Hc) { yo = 0; ++bo; } }   // the next position of the class
          const bool live = orow0 + r < rows;
          oaddr[r] = live ? (((size_t)(img0 + bo) * g.H + (g.sh * yo + py)) * g.W + (g.sw * xo + px)) * g.Cin + j : 0;
#pragma unroll
          for (int cb = 0; cb < CB; ++cb) xgv[r][cb] = xg ? xg[oaddr[r] + 16 * cb] : 0.0f;
        }
      }
      f32x4 acc[CB];
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) acc[cb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      constexpr int G = 8 / NCH;
      f32x4 a[G][NCH];
      int ti = 0, tj = 0;                                        // tap (ti, tj) of t0, advanced incrementally (uniform)
      for (int t0 = 0; t0 < ntap; t0 += G) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
          if (t0 + u >= ntap) break;
          const float* zr = zlane - (ti * pd.Wp + tj) * LDZ;     // dZ[yq - ti][xq - tj], zero outside the output
#pragma unroll
          for (int ch = 0; ch < NCH; ++ch) a[u][ch] = *reinterpret_cast<const f32x4*>(zr + 16 * ch);
          if (++tj == ntj) { tj = 0; ++ti; }
        }
#pragma unroll
        for (int u = 0; u < G; ++u) {
          if (t0 + u >= ntap) break;
          const float* wt = Wc_lds + (t0 + u) * TAP + lane;
#pragma unroll
          for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int cb = 0; cb < CB; ++cb)
                acc[cb] = mfma16(a[u][ch][r], wt[((ch * 4 + r) * CB + cb) * 64], acc[cb]);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (orow0 + r >= rows) break;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
          float v = acc[cb][r];
          if (xg) v *= dx_dact(g.x_gate_act, xgv[r][cb]);       // hand the previous layer its dZ, not its dY
          dx[oaddr[r] + 16 * cb] = v;
        }
      }
    }
  }
}

// What the launch heuristics need to know about the CURRENT device (asked once per device: a process may hold several),
// and the development switches of this file (read once per process).
struct DxDev { int cus, lds; };
static int dx_device() { int d = 0; return hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64 ? d : 0; }
static DxDev dx_dev() {
  static DxDev caps[64];
  static bool known[64];
  const int d = dx_device();
  if (!known[d]) {
    int cus = 0, lds = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || cus <= 0) cus = 256;
    if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, d) != hipSuccess || lds <= 0) lds = 160 * 1024;
    (void)hipGetLastError();
    caps[d] = DxDev{cus, lds};
    known[d] = true;
  }
  return caps[d];
}
static int dx_env(const char* name) {                  // -1: not set
  const char* v = getenv(name);
  return v ? atoi(v) : -1;
}
static const int ENV_CLASS_FORM = dx_env("TRL_DX_CLASS_FORM"), ENV_IMG = dx_env("TRL_DX_IMG"), ENV_WAVES = dx_env("TRL_DX_WAVES");

// images per workgroup of the image-tile form, 0 = the layer does not fit (use the class form): all classes' weights plus
// the padded dZ of the images within 150 KB of LDS; as many images as keeps the grid at one workgroup per CU or more
static int dx_img_per_wg(const DxGeom& g, int* lds_bytes) {
  const int64_t wbytes = (int64_t)g.Cout * g.Cin * g.kh * g.kw * 4;
  const DxPad pd = dx_pad(g);
  const int64_t zbytes = (int64_t)pd.Hp * pd.Wp * (g.Cout + 4) * 4;    // an image's dZ with its zero border
  const DxDev dev = dx_dev();
  const int64_t cap = dev.lds - 10 * 1024;                                   // (150 KB of MI355X's 160)
  if (ENV_CLASS_FORM > 0) return 0;
  if (wbytes + zbytes > cap || (int64_t)pd.Hp * pd.Wp > 4096) return 0;
  // The fewest images per workgroup for which the WHOLE grid is resident at once (160 KB of LDS per CU, 256 CUs): measured
  // at cfg 5 (tools/ab_convdx.py, MI355X): conv 2 (45 KB per image-workgroup: 512 workgroups, two per CU) 29 us against
  // 33 / 51 us with 2 / 4 images; conv 3 (87 KB: one per CU) 27 us with 2 images = 256 workgroups against 35 us with 1
  // image = two rounds and 44 us with 4 = half the CUs idle.
  int img = 1;
  for (; img < 8; ++img) {
    const int64_t lds = wbytes + img * zbytes;
    if (lds + zbytes > cap) break;                                         // one more image would not fit
    const int64_t resident = dev.cus * std::max<int64_t>(1, std::min<int64_t>(8, dev.lds / lds));
    if ((g.B + img - 1) / img <= resident) break;
  }
  if (ENV_IMG >= 0) img = std::max(1, ENV_IMG);
  while (img > 1 && wbytes + img * zbytes > cap) --img;
  img = std::min(img, std::max(1, g.B));
  *lds_bytes = (int)(wbytes + img * zbytes);
  return img;
}
template <int CB, int NCH, int WAVES>
static int launch_dx_img(const float* dy, const float* yg, const float* w, float* wprep, float* dx, const float* xg,
                         const DxGeom& g, int img, int lds, hipStream_t s, bool prepped) {
  static int attr_lds[64];                                                  // the raised dynamic-LDS limit is per device
  const int d = dx_device();
  if (lds > attr_lds[d]) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_dx_img_kernel<CB, NCH, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { trl_set_error("conv_bwd_input: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_lds[d] = lds;
  }
  if (!prepped) {
    const int total = g.Cout * g.Cin * g.kh * g.kw;
    hipLaunchKernelGGL(conv_dx_prep_kernel, dim3(std::min(64, trl_ceil_div(total, 256))), dim3(256), 0, s, w, wprep, g);
    TRL_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL((conv_dx_img_kernel<CB, NCH, WAVES>), dim3(trl_ceil_div(g.B, img)), dim3(64 * WAVES), lds, s, dy, yg, wprep,
                     dx, xg, g, img);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

template <int CB, int NCH, int WAVES>
static int launch_dx_waves(const float* dy, const float* yg, const float* w, float* wprep, float* dx, const float* xg,
                           DxGeom g, int lds, hipStream_t s, bool prepped) {
  static int attr_lds[64];
  const int d = dx_device();
  if (lds > attr_lds[d]) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_dx_kernel<CB, NCH, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { trl_set_error("conv_bwd_input: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_lds[d] = lds;
  }
  const int total = g.Cout * g.Cin * g.kh * g.kw;
  if (!prepped) {
    hipLaunchKernelGGL(conv_dx_prep_kernel, dim3(std::min(64, trl_ceil_div(total, 256))), dim3(256), 0, s, w, wprep, g);
    TRL_LAUNCH_CHECK();
  }
  const int64_t rows_max = (int64_t)g.B * trl_ceil_div(g.H, g.sh) * trl_ceil_div(g.W, g.sw);   // class (0, 0) is the largest
  const int tiles = trl_ceil_div(rows_max, 16 * WAVES), classes = g.sh * g.sw;
  // enough workgroups for ~8 per CU (what hides the loads is waves in flight); beyond that a workgroup walks several
  // tiles on one staging of the weights
  g.tiles_per_wg = std::max(1, std::min(8, (int)((int64_t)tiles * classes / 2048)));
#ifdef TRL_EXP_DX
  g.dbg = getenv("TRL_DX_DBG") ? atoi(getenv("TRL_DX_DBG")) : 0;
  if (getenv("TRL_DX_TPW")) g.tiles_per_wg = atoi(getenv("TRL_DX_TPW"));
#endif
  hipLaunchKernelGGL((conv_dx_kernel<CB, NCH, WAVES>), dim3(trl_ceil_div(tiles, g.tiles_per_wg), classes), dim3(64 * WAVES),
                     lds, s, dy, yg, wprep, dx, xg, g);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}
template <int CB, int NCH>
static int launch_dx(const float* dy, const float* yg, const float* w, float* wprep, float* dx, const float* xg,
                     const DxGeom& g, hipStream_t s, bool prepped) {
  {
    int lds_img = 0;
    const int img = dx_img_per_wg(g, &lds_img);
    if (img > 0) {
      // eight waves (two per SIMD) hide the LDS round trips of the tap walk: faster than four at every measured
      // geometry (29 vs 34 us, 27 vs 34 us); a workgroup with at most 4 blocks of 16 positions has nothing for them to do
      int blocks = 0;
      for (int cls = 0; cls < g.sh * g.sw; ++cls)
        blocks += trl_ceil_div(img * trl_ceil_div(g.H - cls / g.sw, g.sh) * trl_ceil_div(g.W - cls % g.sw, g.sw), 16);
      bool eight = blocks > 4;
      if (ENV_WAVES >= 0) eight = ENV_WAVES >= 8;
      return eight ? launch_dx_img<CB, NCH, 8>(dy, yg, w, wprep, dx, xg, g, img, lds_img, s, prepped)
                   : launch_dx_img<CB, NCH, 4>(dy, yg, w, wprep, dx, xg, g, img, lds_img, s, prepped);
    }
  }
  const int max_taps = trl_ceil_div(g.kh, g.sh) * trl_ceil_div(g.kw, g.sw);
  const int lds = max_taps * g.Cout * g.Cin * (int)sizeof(float);
  TRL_REQUIRE(lds <= dx_dev().lds, "conv_bwd_input: one parity class of the weights exceeds the LDS");
  // a class's weights above ~40 KB leave room for two or three workgroups per CU: make them 8 waves each
  if (lds > 40 * 1024) return launch_dx_waves<CB, NCH, 8>(dy, yg, w, wprep, dx, xg, g, lds, s, prepped);
  return launch_dx_waves<CB, NCH, 4>(dy, yg, w, wprep, dx, xg, g, lds, s, prepped);
}

template <int CB>
static int launch_dx_cout(const float* dy, const float* yg, const float* w, float* wprep, float* dx, const float* xg,
                          const DxGeom& g, hipStream_t s, bool prepped) {
  switch (g.Cout >> 4) {
    case 1: return launch_dx<CB, 1>(dy, yg, w, wprep, dx, xg, g, s, prepped);
    case 2: return launch_dx<CB, 2>(dy, yg, w, wprep, dx, xg, g, s, prepped);
    default: return launch_dx<CB, 4>(dy, yg, w, wprep, dx, xg, g, s, prepped);
  }
}

extern "C" int trl_conv_bwd_input_nhwc_ok(int Cin, int Cout, int kh, int kw, int sh, int sw) {
  if (Cin <= 0 || (Cin & 15) || Cin > 64 || (Cout != 16 && Cout != 32 && Cout != 64)) return 0;
  if (kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || sh > kh || sw > kw) return 0;
  return (int64_t)trl_ceil_div(kh, sh) * trl_ceil_div(kw, sw) * Cout * Cin * 4 <= 160 * 1024;
}

extern "C" int trl_conv_bwd_input_nhwc_workspace(int Cin, int Cout, int kh, int kw) {
  if (Cin <= 0 || Cout <= 0 || kh <= 0 || kw <= 0) return 0;
  return Cin * Cout * kh * kw;                       // floats: the re-ordered weights
}

extern "C" int trl_conv_bwd_input_nhwc_prep_f32(int n, const float* const* w, float* const* workspace, const int* Cin,
                                                const int* Cout, const int* kh, const int* kw, const int* sh, const int* sw,
                                                void* stream) {
  TRL_REQUIRE(n >= 1 && n <= DX_PREP_MAX, "conv_bwd_input prep: 1..8 layers per launch");
  TRL_REQUIRE(w && workspace && Cin && Cout && kh && kw && sh && sw, "null pointer array");
  DxPrepSet p{};
  int most = 0;
  for (int i = 0; i < n; ++i) {
    TRL_REQUIRE(w[i] && workspace[i] && trl_conv_bwd_input_nhwc_ok(Cin[i], Cout[i], kh[i], kw[i], sh[i], sw[i]),
                "conv_bwd_input prep: null pointer / geometry outside trl_conv_bwd_input_nhwc_ok");
    p.w[i] = w[i]; p.wprep[i] = workspace[i];
    p.g[i] = DxGeom{0, Cin[i], 0, 0, kh[i], kw[i], sh[i], sw[i], 0, 0, Cout[i], TRL_ACT_NONE, TRL_ACT_NONE, 1};
    most = std::max(most, Cout[i] * Cin[i] * kh[i] * kw[i]);
  }
  hipLaunchKernelGGL(conv_dx_prep_multi_kernel, dim3(std::min(64, trl_ceil_div(most, 256)), n), dim3(256), 0, (hipStream_t)stream, p);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

extern "C" int trl_conv_bwd_input_nhwc_f32(const float* dy, const float* y_gate, int gate_act, const float* w, float* dx,
                                           const float* x_gate, int x_gate_act, float* workspace, int B, int Cin, int H,
                                           int W, int kh, int kw, int sh, int sw, int Cout, int prepped, void* stream) {
  TRL_REQUIRE(B >= 0 && H >= kh && W >= kw, "bad geometry");
  TRL_REQUIRE(trl_conv_bwd_input_nhwc_ok(Cin, Cout, kh, kw, sh, sw),
              "needs Cin a multiple of 16 (<= 64), Cout 16 / 32 / 64, stride <= kernel (else trl_linear_bwd_input_f32 + trl_col2im_f32)");
  if (B == 0) return TRL_OK;
  TRL_REQUIRE(dy && w && dx && workspace, "null pointer");
  TRL_REQUIRE(gate_act == TRL_ACT_TANH || gate_act == TRL_ACT_RELU || gate_act == TRL_ACT_NONE, "unknown activation");
  TRL_REQUIRE(x_gate_act == TRL_ACT_TANH || x_gate_act == TRL_ACT_RELU || x_gate_act == TRL_ACT_NONE, "unknown activation");
  TRL_REQUIRE((reinterpret_cast<uintptr_t>(dy) & 15) == 0 && (reinterpret_cast<uintptr_t>(y_gate) & 15) == 0 &&
              (reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "dy / y_gate / workspace must be 16-byte aligned");
  TRL_REQUIRE((int64_t)B * H * W * Cin < ((int64_t)1 << 31) && (int64_t)B * H * W < ((int64_t)1 << 23), "tensor too large");
  DxGeom g{B, Cin, H, W, kh, kw, sh, sw, (H - kh) / sh + 1, (W - kw) / sw + 1, Cout, gate_act, x_gate_act, 1};
  hipStream_t s = (hipStream_t)stream;
  switch (Cin >> 4) {
    case 1: return launch_dx_cout<1>(dy, y_gate, w, workspace, dx, x_gate, g, s, prepped != 0);
    case 2: return launch_dx_cout<2>(dy, y_gate, w, workspace, dx, x_gate, g, s, prepped != 0);
    case 3: return launch_dx_cout<3>(dy, y_gate, w, workspace, dx, x_gate, g, s, prepped != 0);
    default: return launch_dx_cout<4>(dy, y_gate, w, workspace, dx, x_gate, g, s, prepped != 0);
  }
}
