// Cross-rank exchange over peer-mapped (hipIpc / xGMI) buffers: device-side helpers shared by k_comm.hip (the
// stand-alone small all-reduce) and k_ppo.hip (fold -> all-reduce -> clip -> Adam in one launch).
//
// The reference has no distributed path; this is the C1 / C2 / C3 exchange of SURVEY.md 8(e) in its
// latency form.  Transport = 8-byte GRANULES {value bits, epoch tag} written by ONE system-scope store each
// (the data is the flag -- no separate barrier, no fence): rank r PUSHES its granules into slot r of EVERY rank's
// buffer (posted xGMI writes), then polls its OWN buffer until all `world` slots carry this call's epoch and sums
// them in rank order, so every rank computes bit-identical totals.  Buffers are uncached device memory
// (hipDeviceMallocUncached) so the owner's polls and the peers' stores both bypass L2.
// Two halves per region, selected by epoch parity: a rank can only be one call ahead of the slowest rank (it needs
// that rank's granules to finish its own call), so a slot is never overwritten while its previous content is being read.
#pragma once
#include "trl_common.h"

#define TRL_MAX_RANKS 16
#define TRL_XR_CAP_GRAD 12288                      // granules per (half, slot) of the gradient region
#define TRL_XR_CAP_SMALL 4096                      // 32-bit words per (half, slot) of the small-message region

struct XrArgs {                                    // passed by value to kernels
  int rank, world;
  unsigned long long* peer[TRL_MAX_RANKS];         // base of every rank's buffer as mapped into THIS process
  unsigned* ctl;                                   // local control words: [0] small-region epochs done, [1] ticket, [2] error,
                                                   //                      [4] gradient-region epochs done, [5] its block ticket,
                                                   //                      [8..10] first time-out: where + 1, epoch, tag found
  unsigned long long wait_ticks;                   // bound of a peer wait in 100 MHz wall-clock ticks (TRL_COMM_TIMEOUT_S, 20 s)
};

struct trl_comm;
const XrArgs* trl_comm_xr(const trl_comm* c);      // device-side view of a communicator whose peers are mapped, else null
int trl_comm_wait_blocks(const trl_comm* c);       // trl_comm_set_wait_footprint's value (0: the kernel's own grid)

// granule offsets inside a rank's buffer
__host__ __device__ inline size_t xr_grad_off(int world, unsigned epoch, int slot, int i) {
  return ((size_t)(epoch & 1u) * world + slot) * TRL_XR_CAP_GRAD + i;
}
__host__ __device__ inline size_t xr_small_base(int world) { return (size_t)2 * world * TRL_XR_CAP_GRAD; }
__host__ __device__ inline size_t xr_small_off(int world, unsigned epoch, int slot, int i) {
  return xr_small_base(world) + ((size_t)(epoch & 1u) * world + slot) * TRL_XR_CAP_SMALL + i;
}
__host__ __device__ inline size_t xr_buffer_granules(int world) {
  return xr_small_base(world) + (size_t)2 * world * TRL_XR_CAP_SMALL;
}

__device__ __forceinline__ void xr_store(unsigned long long* p, unsigned epoch, unsigned bits) {
  __hip_atomic_store(p, ((unsigned long long)epoch << 32) | bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// polls one local granule until its tag is `epoch`; bounded by wall-clock time (100 MHz counter; `ticks`, 20 s unless the
// communicator was created under another TRL_COMM_TIMEOUT_S) so that a missing rank trips ctl[2] instead of hanging the GPU.
// A caller that goes on to WRITE state (the Adam step) must look at ctl[2] first: the sum is partial after a time-out.
// `where` = (region << 8) | slot of the granule (region 1: gradient, 2: statistics): the first wait that times out leaves
// {where + 1, epoch, tag found} in ctl[8..10] for trl_comm_error_detail -- which rank's contribution was missing.
__device__ __forceinline__ unsigned xr_wait(unsigned long long* p, unsigned epoch, unsigned* ctl,
                                            unsigned long long ticks = 2000000000ull, unsigned where = 0u) {
  unsigned long long v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if ((unsigned)(v >> 32) == epoch) return (unsigned)v;
  const unsigned long long t0 = wall_clock64();
  for (unsigned it = 1;; ++it) {
    __builtin_amdgcn_s_sleep(2);
    v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((unsigned)(v >> 32) == epoch) return (unsigned)v;
    if ((it & 1023u) == 0 && wall_clock64() - t0 > ticks) {
      __hip_atomic_store(ctl + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned expect = 0u;
      if (__hip_atomic_compare_exchange_strong(ctl + 8, &expect, where + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_store(ctl + 9, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ctl + 10, (unsigned)(v >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return 0u;
    }
  }
}

// every lane: push `bits` into granule i of my slot on every rank, then collect granule i of all slots (rank order)
__device__ __forceinline__ float xr_allsum_f32(const XrArgs& x, unsigned epoch, int i, float v, bool active) {
  if (!active) return 0.0f;
  const unsigned bits = __float_as_uint(v);
  for (int p = 0; p < x.world; ++p) xr_store(x.peer[p] + xr_grad_off(x.world, epoch, x.rank, i), epoch, bits);
  float s = 0.0f;
  unsigned long long* mine = x.peer[x.rank];
  for (int q = 0; q < x.world; ++q)
    s += __uint_as_float(xr_wait(mine + xr_grad_off(x.world, epoch, q, i), epoch, x.ctl, x.wait_ticks, 0x100u | (unsigned)q));
  return s;
}
