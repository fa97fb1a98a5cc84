// Host-side glue: thread-local error string + ABI version.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include "../../include/trl_hip.h"

static thread_local char g_err[512] = "";

void trl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* trl_last_error(void) { return g_err; }
extern "C" int trl_abi_version(void) { return 1; }

// ---- MT19937 state advance (host) ---------------------------------------------------------------------------------
// The reference draws its exploration noise from the CPU torch generator (torch/policies/distribution.py:60-76), whose
// engine is MT19937 (ATen/core/MT19937RNGEngine.h): a call does `if (--left == 0) next_state(); y = state[next++];` and a
// float32 normal_() of n >= 16 elements (n % 16 == 0) makes exactly n calls.  Advancing the STATE by n calls without
// producing outputs costs one in-place twist of the 624 words per 624 calls (three loops with dependency distances 397 /
// 227, so the compiler vectorises them): the state after any prefix of a large draw can be handed to another generator,
// and P host threads then produce the P segments of ONE torch.randn block concurrently -- the same values in the same
// places, bit for bit (torchrl_amd/collector/noise.py; tests/test_host_logic_cpu.py).
static inline uint32_t mt_twist(uint32_t u, uint32_t v) {
  return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}
static void mt_next_state(uint32_t* st) {
  enum { N = 624, M = 397 };
  uint32_t first = st[0];
  for (int i = 0; i < N - M; ++i) st[i] = st[i + M] ^ mt_twist(st[i], st[i + 1]);          // reads old words only
  for (int i = N - M; i < N - 1; ++i) st[i] = st[i + M - N] ^ mt_twist(st[i], st[i + 1]);  // st[i - 227]: new words
  (void)first;
  st[N - 1] = st[M - 1] ^ mt_twist(st[N - 1], st[0]);                                      // wraps to the NEW st[0]
}
extern "C" int trl_mt19937_advance(uint32_t* state, int32_t* left, int64_t* next, int64_t calls) {
  if (!state || !left || !next || calls < 0 || *left < 1 || *left > 624 || *next < 0 || *next > 624) {
    trl_set_error("trl_mt19937_advance: bad state (left %d, next %lld, calls %lld)", left ? *left : -1,
                  next ? (long long)*next : -1ll, (long long)calls);
    return TRL_EINVAL;
  }
  int64_t k = calls;
  int lf = *left;
  int64_t nx = *next;
  while (k > 0) {
    const int64_t avail = lf - 1;                 // calls that do not regenerate
    if (k <= avail) { lf -= (int)k; nx += k; k = 0; break; }
    k -= avail;                                   // (left is 1 now: the next call regenerates and consumes word 0)
    mt_next_state(state);
    lf = 624; nx = 1; k -= 1;
  }
  *left = lf; *next = nx;
  return TRL_OK;
}

// K engine states of ONE stream in one call: record k = the generator-state image `tmpl` (state_bytes bytes; engine fields at
// off_left (int32) / off_next (int64) / off_mt (624 x uint64, torch's CPUGeneratorImplState)) moved forward to engine call
// pos[k] (ascending, relative to tmpl).  With env shards on several ranks, rank r needs the states at t * N_total * A +
// r * N_local * A for every step t of a rollout (its rows of each step's (N_total, A) draw, distribution.py:60-76) plus the
// state at the end of the block: one pass over the stream, no Python per record.
extern "C" int trl_mt19937_states_at(const uint8_t* tmpl, int64_t state_bytes, int64_t off_left, int64_t off_next,
                                     int64_t off_mt, const int64_t* pos, int64_t K, uint8_t* out) {
  if (!tmpl || !pos || !out || K < 0 || off_left < 0 || off_next < 0 || off_mt < 0 || off_left + 4 > state_bytes ||
      off_next + 8 > state_bytes || off_mt + 624 * 8 > state_bytes) {
    trl_set_error("trl_mt19937_states_at: bad arguments (state_bytes %lld, K %lld)", (long long)state_bytes, (long long)K);
    return TRL_EINVAL;
  }
  uint32_t st[624];
  int32_t left;
  int64_t next;
  memcpy(&left, tmpl + off_left, 4);
  memcpy(&next, tmpl + off_next, 8);
  for (int i = 0; i < 624; ++i) { uint64_t w; memcpy(&w, tmpl + off_mt + 8 * i, 8); st[i] = (uint32_t)w; }
  int64_t at = 0;
  for (int64_t k = 0; k < K; ++k) {
    if (pos[k] < at) { trl_set_error("trl_mt19937_states_at: positions must ascend (pos[%lld])", (long long)k); return TRL_EINVAL; }
    int rc = trl_mt19937_advance(st, &left, &next, pos[k] - at);
    if (rc != TRL_OK) return rc;
    at = pos[k];
    uint8_t* rec = out + k * state_bytes;
    memcpy(rec, tmpl, (size_t)state_bytes);
    memcpy(rec + off_left, &left, 4);
    memcpy(rec + off_next, &next, 8);
    for (int i = 0; i < 624; ++i) { uint64_t w = st[i]; memcpy(rec + off_mt + 8 * i, &w, 8); }
  }
  return TRL_OK;
}
