// MT19937 jump-ahead (host): the K engine states of trl_mt19937_states_at, derived by P threads at once.
//
// Why: with env shards on W ranks every rank draws only ITS rows of each step's (N_total, A) noise tensor
// (torchrl/policies/distribution.py:60-76 draws the tensor for all envs), but to know the engine state at its T chunk
// starts it has to walk the WHOLE stream of the rollout -- T * N_total * A engine calls, 12.6 M at BASELINE cfg 4, 2.4 ms of
// one host thread per rollout and rank however many threads draw the values afterwards.  The walk is sequential only as
// long as the state is moved by stepping it: MT19937's transition F is linear over GF(2) on its 624-word window, so
// F^J = g_J(F) with g_J = x^J mod phi, phi the (degree-19937) minimal polynomial of the recurrence, and g_J(F) applied to a
// window costs ~20 k single steps plus ~2.5 k window XORs whatever J is (Haramoto, Matsumoto, Nishimura, Panneton,
// L'Ecuyer, "Efficient jump ahead for F2-linear random number generators", 2008 -- restated here, no code of theirs).
// So the position list is cut into P contiguous groups; group p > 0 starts from the template JUMPED to a block boundary
// shortly before its first position and walks its own part.  J depends only on the (fixed) relative positions, so every
// polynomial is computed once per process (~0.1 s: square-and-multiply mod phi) and cached.
//
// Exactness.  The window (x_k .. x_{k+623}) carries 31 dead bits (the low bits of x_k: the recurrence reads only its top
// bit) on which g(F) and F^J may differ -- and torch's state image holds them.  Every jump therefore lands ONE block early
// and regenerates once: all 624 words of the block that is handed on are freshly computed from live bits, i.e. the state
// images returned are byte-identical to the ones the sequential walk returns (tests/test_host_logic_cpu.py).
#include <stdint.h>
#include <string.h>
#include <map>
#include <mutex>
#include <thread>
#include <vector>
#include "../../include/trl_hip.h"

void trl_set_error(const char* fmt, ...);

namespace {
enum { MT_N = 624, MT_M = 397, DEG = 19937, PW = 314, PW2 = 2 * PW };   // PW words hold degrees 0 .. 20095

inline uint32_t twist(uint32_t u, uint32_t v) {
  return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}
void next_block(uint32_t* st) {                    // the whole block regenerated in place (ATen's mt19937::next_state)
  for (int i = 0; i < MT_N - MT_M; ++i) st[i] = st[i + MT_M] ^ twist(st[i], st[i + 1]);
  for (int i = MT_N - MT_M; i < MT_N - 1; ++i) st[i] = st[i + MT_M - MT_N] ^ twist(st[i], st[i + 1]);
  st[MT_N - 1] = st[MT_M - 1] ^ twist(st[MT_N - 1], st[0]);
}

struct Poly { uint64_t w[PW]; };
inline int bit(const uint64_t* p, int i) { return (int)((p[i >> 6] >> (i & 63)) & 1u); }

// ---- phi: Berlekamp-Massey over GF(2) on one output bit of the recurrence ----
Poly g_phi;                                         // phi(x), degree DEG
std::vector<uint64_t> g_phi_sh;                     // phi << s for s = 0 .. 63, (PW + 1) words each
std::once_flag g_phi_once;
bool g_phi_ok = false;

void xor_shl(uint64_t* dst, const uint64_t* src, int shift, int nw_src, int nw_dst) {   // dst ^= src << shift
  const int ws = shift >> 6, bs = shift & 63;
  for (int k = 0; k < nw_src; ++k) {
    if (k + ws < nw_dst) dst[k + ws] ^= src[k] << bs;
    if (bs && k + ws + 1 < nw_dst) dst[k + ws + 1] ^= src[k] >> (64 - bs);
  }
}

void phi_init() {
  const int NB = 2 * DEG + 128, BW = (DEG + 64) / 64 + 2;           // sequence bits; words of C / B / the reversed window
  std::vector<uint8_t> s(NB);
  {
    uint32_t st[MT_N];
    st[0] = 5489u;                                                    // any non-zero state: phi is irreducible
    for (int i = 1; i < MT_N; ++i) st[i] = 1812433253u * (st[i - 1] ^ (st[i - 1] >> 30)) + (uint32_t)i;
    for (int n = 0; n < NB;) {
      next_block(st);
      for (int i = 0; i < MT_N && n < NB; ++i, ++n) s[n] = (uint8_t)(st[i] & 1u);
    }
  }
  std::vector<uint64_t> C(BW, 0), B(BW, 0), T(BW, 0), R(BW, 0);
  C[0] = B[0] = 1;
  int L = 0, m = 1;
  for (int n = 0; n < NB; ++n) {
    for (int k = BW - 1; k > 0; --k) R[k] = (R[k] << 1) | (R[k - 1] >> 63);       // R[i] = s[n - i]
    R[0] = (R[0] << 1) | s[n];
    uint64_t acc = 0;
    const int lw = (L >> 6) + 1;
    for (int k = 0; k < lw && k < BW; ++k) acc ^= C[k] & R[k];
    if ((__builtin_popcountll(acc) & 1) == 0) { ++m; continue; }
    if (2 * L <= n) {
      T = C;
      xor_shl(C.data(), B.data(), m, BW, BW);
      L = n + 1 - L;
      B = T;
      m = 1;
    } else {
      xor_shl(C.data(), B.data(), m, BW, BW);
      ++m;
    }
  }
  if (L != DEG) return;                                               // (cannot happen for MT19937; g_phi_ok stays false)
  memset(&g_phi, 0, sizeof(g_phi));
  for (int i = 0; i <= DEG; ++i)                                      // connection polynomial reversed: phi_j = C_{DEG - j}
    if (bit(C.data(), i)) g_phi.w[(DEG - i) >> 6] |= 1ull << ((DEG - i) & 63);
  g_phi_sh.assign((size_t)64 * (PW + 1), 0);
  for (int sft = 0; sft < 64; ++sft) xor_shl(&g_phi_sh[(size_t)sft * (PW + 1)], g_phi.w, sft, PW, PW + 1);
  g_phi_ok = bit(g_phi.w, DEG) && bit(g_phi.w, 0);
}

// acc (PW2 words, degree < 2 * DEG) reduced mod phi into out
void reduce(uint64_t* acc, Poly& out) {
  for (int i = 2 * DEG; i >= DEG; --i) {
    if (!bit(acc, i)) continue;
    const int d = i - DEG, ws = d >> 6;
    const uint64_t* ph = &g_phi_sh[(size_t)(d & 63) * (PW + 1)];
    for (int k = 0; k <= PW && ws + k < PW2; ++k) acc[ws + k] ^= ph[k];
  }
  memcpy(out.w, acc, sizeof(out.w));
}

void mulmod(const Poly& a, const Poly& b, Poly& out) {
  std::vector<uint64_t> ash((size_t)64 * (PW + 1), 0), acc(PW2, 0);
  for (int sft = 0; sft < 64; ++sft) xor_shl(&ash[(size_t)sft * (PW + 1)], a.w, sft, PW, PW + 1);
  for (int i = 0; i < DEG; ++i) {
    if (!bit(b.w, i)) continue;
    const uint64_t* av = &ash[(size_t)(i & 63) * (PW + 1)];
    uint64_t* dst = acc.data() + (i >> 6);
    for (int k = 0; k <= PW && (i >> 6) + k < PW2; ++k) dst[k] ^= av[k];
  }
  reduce(acc.data(), out);
}

void pow_x(uint64_t J, Poly& out) {                                   // x^J mod phi
  Poly r;
  memset(&r, 0, sizeof(r));
  r.w[0] = 1;
  int top = 63;
  while (top >= 0 && !((J >> top) & 1ull)) --top;
  for (int b = top; b >= 0; --b) {
    Poly sq;
    mulmod(r, r, sq);
    r = sq;
    if ((J >> b) & 1ull) {                                            // r *= x
      uint64_t carry = 0;
      for (int k = 0; k < PW; ++k) { const uint64_t nc = r.w[k] >> 63; r.w[k] = (r.w[k] << 1) | carry; carry = nc; }
      if (bit(r.w, DEG)) for (int k = 0; k < PW; ++k) r.w[k] ^= g_phi.w[k];
    }
  }
  out = r;
}

struct CachedPoly { Poly p; uint64_t used; };
std::map<uint64_t, CachedPoly> g_cache;
std::mutex g_cache_mutex;
uint64_t g_cache_clock = 0;
Poly jump_poly(uint64_t J) {                                          // by value: the cache may drop entries under other threads
  {
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    auto it = g_cache.find(J);
    if (it != g_cache.end()) { it->second.used = ++g_cache_clock; return it->second.p; }
  }
  Poly p;
  pow_x(J, p);                                                        // outside the lock: several threads may compute their own
  std::lock_guard<std::mutex> lock(g_cache_mutex);
  // bounded at 256 polynomials (~2.5 KB each); the LEAST RECENTLY USED one goes, so that a run whose chunk layout changes
  // now and then keeps the polynomials of the layouts it returns to (a wholesale clear() made every one of them ~0.1 s again)
  while (g_cache.size() >= 256) {
    auto victim = g_cache.begin();
    for (auto it = g_cache.begin(); it != g_cache.end(); ++it) if (it->second.used < victim->second.used) victim = it;
    g_cache.erase(victim);
  }
  g_cache[J] = CachedPoly{p, ++g_cache_clock};
  return p;
}

// ---- g(F) applied to a 624-word window: Horner in F^W with a table of the 2^W - 1 combinations of W shifted copies ----
enum { JW = 8 };
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define TRL_HOST_CLONES __attribute__((target_clones("avx2", "default")))   // (this file also passes through the device compiler)
#else
#define TRL_HOST_CLONES
#endif
TRL_HOST_CLONES void xor_window(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src) {
  for (int m = 0; m < MT_N; ++m) dst[m] ^= src[m];
}
void jump_window(uint32_t* st, const Poly& g) {
  std::vector<uint32_t> S(MT_N + JW), tbl((size_t)(1 << JW) * MT_N, 0);
  memcpy(S.data(), st, MT_N * sizeof(uint32_t));
  for (int j = 0; j < JW; ++j) S[MT_N + j] = S[MT_M + j] ^ twist(S[j], S[j + 1]);
  for (int c = 1; c < (1 << JW); ++c) {                                // tbl[c] = XOR over set bits j of c of F^j s
    const int low = __builtin_ctz(c), rest = c & (c - 1);
    uint32_t* t = &tbl[(size_t)c * MT_N];
    const uint32_t* base = &tbl[(size_t)rest * MT_N];
    for (int m = 0; m < MT_N; ++m) t[m] = base[m] ^ S[m + low];
  }
  const int nch = (DEG + JW - 1) / JW;
  std::vector<uint32_t> T((size_t)MT_N + (size_t)(nch + 1) * JW + 8, 0);
  size_t at = 0;
  bool live = false;
  for (int ch = nch - 1; ch >= 0; --ch) {
    int c = 0;
    for (int j = 0; j < JW; ++j) { const int i = ch * JW + j; if (i < DEG && bit(g.w, i)) c |= 1 << j; }
    if (live) {
      for (int t = 0; t < JW; ++t, ++at) T[at + MT_N] = T[at + MT_M] ^ twist(T[at], T[at + 1]);
    }
    if (c) { xor_window(&T[at], &tbl[(size_t)c * MT_N]); live = true; }
  }
  memcpy(st, &T[at], MT_N * sizeof(uint32_t));
}

// walks (array = a block with `consumed` words already handed out, 0 .. 624) forward by `calls` engine calls
void walk(uint32_t* st, int64_t& consumed, int64_t calls) {
  while (calls > 0) {
    const int64_t avail = MT_N - consumed;
    if (calls <= avail) { consumed += calls; return; }
    calls -= avail;
    next_block(st);
    consumed = 0;
  }
}
}  // namespace

extern "C" int trl_mt19937_jump_ready(void) {
  std::call_once(g_phi_once, phi_init);
  return g_phi_ok ? 1 : 0;
}

// trl_mt19937_states_at by `threads` host threads (same arguments, same records, byte for byte).  Positions are cut into
// `threads` contiguous groups of about equal stream length; groups that start at least `min_jump` calls into the stream start
// from a jumped state.  threads <= 1 or a short stream: the sequential pass.
extern "C" int trl_mt19937_states_at_mt(const uint8_t* tmpl, int64_t state_bytes, int64_t off_left, int64_t off_next,
                                        int64_t off_mt, const int64_t* pos, int64_t K, uint8_t* out, int threads) {
  const int64_t min_jump = 1 << 18;
  if (threads <= 1 || K < 2 || !pos || pos[K - 1] < 2 * min_jump || !trl_mt19937_jump_ready())
    return trl_mt19937_states_at(tmpl, state_bytes, off_left, off_next, off_mt, pos, K, out);
  if (!tmpl || !out || off_left < 0 || off_next < 0 || off_mt < 0 || off_left + 4 > state_bytes ||
      off_next + 8 > state_bytes || off_mt + 624 * 8 > state_bytes) {
    trl_set_error("trl_mt19937_states_at_mt: bad arguments (state_bytes %lld, K %lld)", (long long)state_bytes, (long long)K);
    return TRL_EINVAL;
  }
  int32_t left0;
  memcpy(&left0, tmpl + off_left, 4);
  if (left0 < 1 || left0 > MT_N) { trl_set_error("trl_mt19937_states_at_mt: bad state (left %d)", left0); return TRL_EINVAL; }
  for (int64_t k = 1; k < K; ++k)
    if (pos[k] < pos[k - 1]) { trl_set_error("trl_mt19937_states_at_mt: positions must ascend (pos[%lld])", (long long)k); return TRL_EINVAL; }
  if (pos[0] < 0) { trl_set_error("trl_mt19937_states_at_mt: negative position"); return TRL_EINVAL; }
  uint32_t st0[MT_N];
  for (int i = 0; i < MT_N; ++i) { uint64_t w; memcpy(&w, tmpl + off_mt + 8 * i, 8); st0[i] = (uint32_t)w; }
  const int64_t c0 = 625 - left0;                                     // words of the template's block already handed out (1 .. 624)
  // group boundaries by stream position
  const int P = (int)(threads > 64 ? 64 : threads);
  std::vector<int64_t> first(P + 1, K);
  first[0] = 0;
  {
    int p = 1;
    for (int64_t k = 0; k < K && p < P; ++k)
      while (p < P && pos[k] >= pos[K - 1] / P * p) first[p++] = k;
  }
  auto run = [&](int p) {
    const int64_t k0 = first[p], k1 = first[p + 1];
    if (k0 >= k1) return;
    uint32_t st[MT_N];
    memcpy(st, st0, sizeof(st));
    int64_t consumed = c0, at = 0;                                    // `at`: stream position (relative to tmpl) of the walker
    // target word index (from the start of the template's block) of pos[k0] is c0 + pos[k0]; land at the start of block
    // Bp = pos[k0] / 624 - 1 (always strictly before it, whatever c0 is): jump the window to block Bp - 1, regenerate once
    const int64_t Bp = pos[k0] / MT_N - 1;
    if (p > 0 && Bp >= 1 && pos[k0] >= min_jump) {
      jump_window(st, jump_poly((uint64_t)MT_N * (uint64_t)(Bp - 1)));
      next_block(st);                                                 // block Bp, every word freshly computed
      consumed = 0;
      at = (int64_t)MT_N * Bp - c0;                                    // stream position of "block Bp, nothing handed out"
    }
    for (int64_t k = k0; k < k1; ++k) {
      walk(st, consumed, pos[k] - at);
      at = pos[k];
      // torch's fields: `next` = words handed out of the current block, `left` = 625 - next (a block with nothing handed
      // out cannot occur here: every position is at least one call past a block start the walker regenerated)
      const int32_t left = (int32_t)(625 - consumed);
      const int64_t next = consumed;
      uint8_t* rec = out + k * state_bytes;
      memcpy(rec, tmpl, (size_t)state_bytes);
      if (!(p == 0 && pos[k] == 0)) {                                 // position 0 = the template itself, fields untouched
        memcpy(rec + off_left, &left, 4);
        memcpy(rec + off_next, &next, 8);
      }
      for (int i = 0; i < MT_N; ++i) { const uint64_t w = st[i]; memcpy(rec + off_mt + 8 * i, &w, 8); }
    }
  };
  // A thread that cannot be created (std::system_error: thread limit of the container, EAGAIN) must not escape through the
  // C ABI -- it would end the process; the groups it would have taken run inline instead.
  std::vector<std::thread> pool;
  int started = 1;
  for (; started < P; ++started) {
    try { pool.emplace_back(run, started); }
    catch (const std::exception&) { break; }
  }
  run(0);
  for (int p = started; p < P; ++p) run(p);
  for (auto& t : pool) t.join();
  return TRL_OK;
}
