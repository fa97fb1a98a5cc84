// C1 / C2 / C3 -- the collectives of the multi-GPU path behind the C ABI.
//
// The reference has no distributed backend; these entry points are what its update loop would call between
// `loss.backward()` and `clip_grad_norm_` (torchrl/algo/on_policy/ppo.py:72-74, 117-119: the clip must see the
// reduced gradient) and around the advantage statistics (ppo.py:141-147).
//
// Two transports behind one `trl_comm_t`:
//   * RCCL (ncclAllReduce on the caller's stream, in place) -- bandwidth-class messages (the 6.6 MB conv gradient) and
//     the fallback for everything; loaded with dlopen so the library itself has no link-time RCCL dependency;
//   * peer-mapped granule exchange (trl_comm.h) -- latency-class messages: the 44 KB PPO gradient (fused into
//     trl_ppo_reduce_adam_xrank_f32, k_ppo.hip) and the few-KB statistics vectors (trl_allreduce_*_f64 here): one
//     kernel, one xGMI hop, no rendezvous kernel, graph-capturable like any other launch.
// The peer buffer is the one allocation the library owns (it must be exported with hipIpcGetMemHandle); everything else
// keeps the no-allocation contract of include/trl_hip.h.
#include <dlfcn.h>
#include <string.h>
#include "trl_comm.h"

typedef struct { char internal[128]; } trl_nccl_uid;                   // ncclUniqueId
typedef void* trl_nccl_comm;
struct RcclApi {
  void* handle;
  int (*GetUniqueId)(trl_nccl_uid*);
  int (*CommInitRank)(trl_nccl_comm*, int, trl_nccl_uid, int);
  int (*CommDestroy)(trl_nccl_comm);
  int (*AllReduce)(const void*, void*, size_t, int, int, trl_nccl_comm, hipStream_t);
  const char* (*GetErrorString)(int);
};
static RcclApi g_rccl = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};

static int load_rccl() {
  if (g_rccl.handle) return TRL_OK;
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) { trl_set_error("trl_comm: cannot load librccl.so.1: %s", dlerror()); return TRL_EUNSUPPORTED; }
  g_rccl.GetUniqueId = (int (*)(trl_nccl_uid*))dlsym(h, "ncclGetUniqueId");
  g_rccl.CommInitRank = (int (*)(trl_nccl_comm*, int, trl_nccl_uid, int))dlsym(h, "ncclCommInitRank");
  g_rccl.CommDestroy = (int (*)(trl_nccl_comm))dlsym(h, "ncclCommDestroy");
  g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, trl_nccl_comm, hipStream_t))dlsym(h, "ncclAllReduce");
  g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce) {
    trl_set_error("trl_comm: librccl lacks a required symbol");
    return TRL_EUNSUPPORTED;
  }
  g_rccl.handle = h;
  return TRL_OK;
}

struct trl_comm {
  int rank, world;
  trl_nccl_comm rccl;                              // null: no RCCL communicator (peer transport only)
  unsigned long long* local;                       // this rank's peer buffer (uncached device memory)
  void* opened[TRL_MAX_RANKS];                     // hipIpcOpenMemHandle results (null for self / not opened)
  XrArgs xr;                                       // device-side view; xr.peer[] valid once peers are open
  int peers_ready;
  int wait_blocks;                                 // resident footprint of a launch that waits for other ranks (0: the kernel's own grid)
  int uncached;                                    // the peer buffer is hipDeviceMallocUncached memory (0: plain hipMalloc fallback)
  hipStream_t peek;                                // trl_comm_error_peek: a non-blocking stream of its own (created on first use)
};

#define HIP_TRY(expr)                                                                         \
  do { hipError_t e_ = (expr);                                                                \
       if (e_ != hipSuccess) { trl_set_error("%s: %s: %s", __func__, #expr, hipGetErrorString(e_)); return (int)e_; } } while (0)

extern "C" int trl_comm_unique_id_bytes(void) { return (int)sizeof(trl_nccl_uid); }
extern "C" int trl_comm_peer_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }
extern "C" int trl_comm_max_ranks(void) { return TRL_MAX_RANKS; }

extern "C" int trl_comm_get_unique_id(void* id_out) {
  TRL_REQUIRE(id_out, "null pointer");
  int rc = load_rccl();
  if (rc) return rc;
  trl_nccl_uid id;
  const int e = g_rccl.GetUniqueId(&id);
  if (e) { trl_set_error("ncclGetUniqueId: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "error"); return 1000 + e; }
  memcpy(id_out, &id, sizeof(id));
  return TRL_OK;
}

// unique_id: the bytes rank 0 obtained from trl_comm_get_unique_id, distributed out of band by the caller (the host
// framework's rendezvous); NULL = no RCCL communicator (peer transport only, e.g. several ranks on one device).
extern "C" int trl_comm_init(trl_comm_t** out, int rank, int world, const void* unique_id) {
  TRL_REQUIRE(out, "null pointer");
  TRL_REQUIRE(world >= 1 && world <= TRL_MAX_RANKS && rank >= 0 && rank < world, "need 0 <= rank < world <= 16");
  trl_comm* c = new trl_comm();
  memset(c, 0, sizeof(*c));
  c->rank = rank; c->world = world;
  c->xr.rank = rank; c->xr.world = world;
  {                                               // bound of every peer wait: 20 s of the 100 MHz wall clock unless configured
    const char* t = getenv("TRL_COMM_TIMEOUT_S"); // (a first graph capture, or rank-0-only I/O, on a slow host may need more)
    const double secs = t ? atof(t) : 20.0;
    c->xr.wait_ticks = (unsigned long long)((secs > 0.001 ? secs : 20.0) * 1e8);
  }
  if (unique_id) {
    int rc = load_rccl();
    if (rc) { delete c; return rc; }
    trl_nccl_uid id;
    memcpy(&id, unique_id, sizeof(id));
    const int e = g_rccl.CommInitRank(&c->rccl, world, id, rank);
    if (e) {
      trl_set_error("ncclCommInitRank: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "error");
      delete c;
      return 1000 + e;
    }
  }
  *out = c;
  return TRL_OK;
}

// Allocates this rank's peer buffer and returns its IPC handle (to be all-gathered by the caller).
extern "C" int trl_comm_peer_export(trl_comm_t* c, void* handle_out) {
  TRL_REQUIRE(c && handle_out, "null pointer");
  if (!c->local) {
    const size_t bytes = xr_buffer_granules(c->world) * sizeof(unsigned long long);
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
    c->uncached = e == hipSuccess ? 1 : 0;
    if (e != hipSuccess) {
      // Said out loud: a CACHED buffer polled across physical devices is what the self-check exists to
      // catch -- the operator learns which allocation this communicator got (trl_comm_peer_buffer_kind, bench.py config).
      fprintf(stderr, "[trl_comm] rank %d: hipExtMallocWithFlags(hipDeviceMallocUncached) failed (%s): the peer buffer is plain "
                      "hipMalloc memory; the self-check decides whether the peer transport is used\n", c->rank, hipGetErrorString(e));
      (void)hipGetLastError();
      HIP_TRY(hipMalloc(&p, bytes));
    }
    HIP_TRY(hipMemset(p, 0, bytes));
    void* ctl = nullptr;
    HIP_TRY(hipMalloc(&ctl, 64));
    HIP_TRY(hipMemset(ctl, 0, 64));
    HIP_TRY(hipDeviceSynchronize());
    c->local = (unsigned long long*)p;
    c->xr.ctl = (unsigned*)ctl;
    c->xr.peer[c->rank] = c->local;
  }
  hipIpcMemHandle_t h;
  HIP_TRY(hipIpcGetMemHandle(&h, c->local));
  memcpy(handle_out, &h, sizeof(h));
  return TRL_OK;
}

// handles: world x trl_comm_peer_handle_bytes() bytes in rank order (the own entry is ignored).
extern "C" int trl_comm_peer_open(trl_comm_t* c, const void* handles) {
  TRL_REQUIRE(c && handles && c->local, "export the local buffer first");
  for (int r = 0; r < c->world; ++r) {
    if (r == c->rank || c->opened[r]) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, (const char*)handles + (size_t)r * sizeof(h), sizeof(h));
    void* p = nullptr;
    HIP_TRY(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    c->opened[r] = p;
    c->xr.peer[r] = (unsigned long long*)p;
  }
  c->peers_ready = 1;
  return TRL_OK;
}

extern "C" int trl_comm_peer_ready(const trl_comm_t* c) { return (c && c->peers_ready) ? 1 : 0; }
// 0: stop using the peer transport (e.g. after a failed self-check); all calls then take the RCCL route
extern "C" int trl_comm_peer_enable(trl_comm_t* c, int on) {
  TRL_REQUIRE(c, "null communicator");
  TRL_REQUIRE(!on || c->local, "peers were never mapped");
  c->peers_ready = on ? 1 : 0;
  return TRL_OK;
}
extern "C" int trl_comm_has_rccl(const trl_comm_t* c) { return (c && c->rccl) ? 1 : 0; }

// 1 when a peer wait timed out since the last call (and clears the flag); synchronises the device.
extern "C" int trl_comm_error(trl_comm_t* c) {
  if (!c || !c->xr.ctl) return 0;
  unsigned v = 0;
  if (hipMemcpy(&v, c->xr.ctl + 2, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  if (v) (void)hipMemset(c->xr.ctl + 2, 0, sizeof(v));
  return v ? 1 : 0;
}

// The same flag WITHOUT waiting for the device: a 4-byte read on a stream of the communicator's own that does not synchronise
// with the caller's streams.  For the once-per-iteration check of a training loop whose host runs ahead of the device (the
// kernels of the update being checked have completed -- the caller waited for their statistics -- while the next update's
// are running): trl_comm_error there stalls the host until the device is idle, once per iteration, and the next rollout is
// then launched onto an idle device (round 6: the reason multi-rank runs staged their exploration-noise blocks instead of
// carrying them).  A time-out raised by work still in flight is seen by the next call.
extern "C" int trl_comm_error_peek(trl_comm_t* c) {
  if (!c || !c->xr.ctl) return 0;
  if (!c->peek && hipStreamCreateWithFlags(&c->peek, hipStreamNonBlocking) != hipSuccess) { c->peek = nullptr; return trl_comm_error(c); }
  unsigned v = 0;
  if (hipMemcpyAsync(&v, c->xr.ctl + 2, sizeof(v), hipMemcpyDeviceToHost, c->peek) != hipSuccess) return 1;
  if (hipStreamSynchronize(c->peek) != hipSuccess) return 1;
  if (v) { (void)hipMemsetAsync(c->xr.ctl + 2, 0, sizeof(v), c->peek); (void)hipStreamSynchronize(c->peek); }
  return v ? 1 : 0;
}

// What the first timed-out wait since the last call was waiting for: out[0] = region (1 gradient, 2 statistics; 0 = no
// time-out recorded), out[1] = slot = the rank whose granules were missing, out[2] = epoch waited for, out[3] = epoch tag
// found in the granule.  Clears the record; synchronises the device.
extern "C" int trl_comm_error_detail(trl_comm_t* c, int32_t* out) {
  if (!out) return TRL_EINVAL;
  out[0] = out[1] = out[2] = out[3] = 0;
  if (!c || !c->xr.ctl) return TRL_OK;
  unsigned v[3] = {0, 0, 0};
  if (hipMemcpy(v, c->xr.ctl + 8, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return TRL_EINVAL;
  if (v[0]) {
    out[0] = (int32_t)((v[0] - 1u) >> 8); out[1] = (int32_t)((v[0] - 1u) & 0xffu); out[2] = (int32_t)v[1]; out[3] = (int32_t)v[2];
    (void)hipMemset(c->xr.ctl + 8, 0, sizeof(v));
  }
  return TRL_OK;
}

// Pre-flight of a multi-GPU run (host, no communicator needed): out[3] = {peer access possible from dev_a to dev_b
// (hipDeviceCanAccessPeer), link type (hsa_amd_link_info_type_t: 2 PCIe, 4 xGMI; -1 unknown), hops}.
extern "C" int trl_comm_link_info(int dev_a, int dev_b, int32_t* out) {
  TRL_REQUIRE(out, "null pointer");
  out[0] = 0; out[1] = -1; out[2] = 0;
  int n = 0;
  HIP_TRY(hipGetDeviceCount(&n));
  TRL_REQUIRE(dev_a >= 0 && dev_b >= 0 && dev_a < n && dev_b < n, "device index out of range");
  if (dev_a == dev_b) { out[0] = 1; return TRL_OK; }
  int can = 0;
  HIP_TRY(hipDeviceCanAccessPeer(&can, dev_a, dev_b));
  out[0] = can;
  uint32_t type = 0, hops = 0;
  if (hipExtGetLinkTypeAndHopCount(dev_a, dev_b, &type, &hops) == hipSuccess) { out[1] = (int32_t)type; out[2] = (int32_t)hops; }
  else (void)hipGetLastError();
  return TRL_OK;
}

extern "C" int trl_comm_destroy(trl_comm_t* c) {
  if (!c) return TRL_OK;
  for (int r = 0; r < c->world; ++r)
    if (c->opened[r]) (void)hipIpcCloseMemHandle(c->opened[r]);
  if (c->local) (void)hipFree(c->local);
  if (c->xr.ctl) (void)hipFree(c->xr.ctl);
  if (c->peek) (void)hipStreamDestroy(c->peek);
  if (c->rccl && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->rccl);
  delete c;
  return TRL_OK;
}

// the device-side view, for the fused kernels in other translation units (k_ppo.hip)
const XrArgs* trl_comm_xr(const trl_comm_t* c) { return (c && c->peers_ready) ? &c->xr : nullptr; }
int trl_comm_wait_blocks(const trl_comm_t* c) { return c ? c->wait_blocks : 0; }

// Resident footprint of launches that wait INSIDE a kernel for other ranks (trl_ppo_reduce_adam_xrank_f32: blocks of 8
// waves that stay on the device until every rank has delivered its gradient).  0 = the kernel's own grid (one block per 64
// parameters: 180 for the benchmark shape) -- right for one rank per GPU.  Ranks SHARING a device must leave CUs for each
// other's gradient kernels: the host sets blocks <= CUs / (2 x ranks per device).
extern "C" int trl_comm_set_wait_footprint(trl_comm_t* c, int blocks) {
  TRL_REQUIRE(c && blocks >= 0, "null communicator / negative block count");
  c->wait_blocks = blocks;
  return TRL_OK;
}
// 1: uncached device memory (hipDeviceMallocUncached), 0: plain hipMalloc (fallback), -1: no peer buffer yet
extern "C" int trl_comm_peer_buffer_kind(const trl_comm_t* c) { return (c && c->local) ? c->uncached : -1; }

// ---------------------------------------------------------------- small one-shot all-reduce
// buf[i] <- reduce over ranks of buf[i]; element i is SUMmed, or MAXed when bit (i % period) of max_mask is set.
// WORDS = 32-bit words per element (1: float, 2: double).  Epoch of the call = ctl[0] + 1, advanced by the last
// block to finish (every block has read it by then), so a captured launch replays with no argument change.
template <typename T, int WORDS>
__global__ __launch_bounds__(256) void xr_allreduce_kernel(T* __restrict__ buf, int n, int period, unsigned long long max_mask,
                                                           XrArgs x) {
  const unsigned epoch = __hip_atomic_load(x.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    union { T v; unsigned w[WORDS]; } u;
    u.v = buf[i];
    for (int p = 0; p < x.world; ++p)
#pragma unroll
      for (int k = 0; k < WORDS; ++k)
        xr_store(x.peer[p] + xr_small_off(x.world, epoch, x.rank, WORDS * i + k), epoch, u.w[k]);
    unsigned long long* mine = x.peer[x.rank];
    const bool is_max = (max_mask >> (i % period)) & 1ull;
    T acc = (T)0;
    for (int q = 0; q < x.world; ++q) {
#pragma unroll
      for (int k = 0; k < WORDS; ++k) u.w[k] = xr_wait(mine + xr_small_off(x.world, epoch, q, WORDS * i + k), epoch, x.ctl, x.wait_ticks, 0x200u | (unsigned)q);
      acc = (q == 0) ? u.v : (is_max ? (u.v > acc ? u.v : acc) : acc + u.v);
    }
    buf[i] = acc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned before = __hip_atomic_fetch_add(x.ctl + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (before == gridDim.x - 1) {
      __hip_atomic_store(x.ctl + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(x.ctl, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Mid-size float vectors (up to TRL_XR_CAP_GRAD: the PPO gradient and anything like it) through the GRADIENT region, with
// the same epoch counter (ctl[4]) and slot arithmetic as the exchange inside trl_ppo_reduce_adam_xrank_f32 -- which this
// kernel therefore also exercises in the communicator's self-check.  ctl[5] is its block ticket.  (ctl[6]: the count of
// the value function's exchanges when a rank runs two update chains, k_ppo.hip; a launch that touches every granule, like
// this one, takes the larger count and leaves both at its epoch)
__global__ __launch_bounds__(256) void xr_allreduce_grad_kernel(float* __restrict__ buf, int n, XrArgs x) {
  const unsigned e4 = __hip_atomic_load(x.ctl + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned e6 = __hip_atomic_load(x.ctl + 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned epoch = (e4 > e6 ? e4 : e6) + 1u;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool act = i < n;
  const float v = xr_allsum_f32(x, epoch, act ? i : 0, act ? buf[i] : 0.0f, act);
  if (act) buf[i] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned before = __hip_atomic_fetch_add(x.ctl + 5, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (before == gridDim.x - 1) {
      __hip_atomic_store(x.ctl + 5, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(x.ctl + 6, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(x.ctl + 4, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <typename T, int WORDS>
static int launch_small(T* buf, int64_t n, int period, unsigned long long mask, trl_comm_t* c, hipStream_t s) {
  hipLaunchKernelGGL((xr_allreduce_kernel<T, WORDS>), dim3(trl_ceil_div(n, 256)), dim3(256), 0, s, buf, (int)n, period, mask, c->xr);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// In-place SUM of n floats over all ranks, on `stream` (C1 of SURVEY.md 8(e)).  Latency transport for n up to
// TRL_XR_CAP_GRAD (12 288) floats when the peers are mapped, RCCL ring otherwise.
extern "C" int trl_allreduce_sum_f32(float* buf, int64_t n, trl_comm_t* c, void* stream) {
  TRL_REQUIRE(c && n >= 0, "null communicator / negative size");
  if (n == 0) return TRL_OK;
  TRL_REQUIRE(buf, "null pointer");
  if (c->peers_ready && n <= TRL_XR_CAP_SMALL) return launch_small<float, 1>(buf, n, 1, 0ull, c, (hipStream_t)stream);
  if (c->peers_ready && n <= TRL_XR_CAP_GRAD) {
    hipLaunchKernelGGL(xr_allreduce_grad_kernel, dim3(trl_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, buf, (int)n, c->xr);
    TRL_LAUNCH_CHECK();
    return TRL_OK;
  }
  if (c->rccl) {
    const int e = g_rccl.AllReduce(buf, buf, (size_t)n, /*ncclFloat32*/ 7, /*ncclSum*/ 0, c->rccl, (hipStream_t)stream);
    if (e) { trl_set_error("ncclAllReduce: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "error"); return 1000 + e; }
    return TRL_OK;
  }
  trl_set_error("trl_allreduce_sum_f32: %lld floats exceed the peer transport and no RCCL communicator exists", (long long)n);
  return TRL_EUNSUPPORTED;
}

// In-place reduction of n doubles (C2 / C3: advantage and logging statistics): element i is MAXed when bit
// (i % period) of max_mask is set, SUMmed otherwise.  Peer transport only (2 n <= TRL_XR_CAP_SMALL); a pure SUM
// (max_mask == 0) falls back to RCCL.
extern "C" int trl_allreduce_f64(double* buf, int64_t n, int period, uint64_t max_mask, trl_comm_t* c, void* stream) {
  TRL_REQUIRE(c && n >= 0 && period >= 1 && period <= 64, "bad arguments");
  if (n == 0) return TRL_OK;
  TRL_REQUIRE(buf, "null pointer");
  if (c->peers_ready && 2 * n <= TRL_XR_CAP_SMALL)
    return launch_small<double, 2>(buf, n, period, (unsigned long long)max_mask, c, (hipStream_t)stream);
  if (c->rccl && max_mask == 0) {
    const int e = g_rccl.AllReduce(buf, buf, (size_t)n, /*ncclFloat64*/ 8, /*ncclSum*/ 0, c->rccl, (hipStream_t)stream);
    if (e) { trl_set_error("ncclAllReduce: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "error"); return 1000 + e; }
    return TRL_OK;
  }
  trl_set_error("trl_allreduce_f64: message too large for the peer transport (or peers not mapped)");
  return TRL_EUNSUPPORTED;
}
