// K6b -- frame-deduplicating replay for stacked-frame observations: the GPU counterpart of the reference's
// LazyFrames + MemoryEfficientReplayBuffer (torchrl/env/atari_wrapper.py:142-227,
// torchrl/replay_buffers/memory_efficient_replay_buffer.py:5-33), which keep every 84x84 frame once and
// rebuild the k-stacks when a batch is encoded.
//
// Per env a FRAME STREAM (ring of S slots x HW bytes): a step appends the newest frame of next_obs, an episode
// (re)start appends the C frames of the fresh stack.  Stream positions are monotone int32 counters
// (slot = pos % S); a replay row stores only pos[row][env] = position of the newest frame of obs, so
//     obs_t      = stream[pos - C + 1 .. pos]        next_obs_t = stream[pos - C + 2 .. pos + 1]
// (the step's new frame is appended at pos + 1 before any reset appends the next episode's stack).
// HBM footprint ~ 1 frame per transition instead of 2 C frames (8x for C = 4); the gather reads C + 1
// frames per sample.  All kernels are streaming byte copies (HBM-bound, 16-byte vectors).
#include "trl_common.h"

#define FR_THREADS 256

// n_frames == 1: append channel C-1 of stacks[n]; n_frames == C: append the whole stack (episode start).
// mask (nullable): only envs with mask[n] != 0.
__global__ __launch_bounds__(FR_THREADS) void frame_append_kernel(const uint8_t* __restrict__ stacks,
                                                                  uint8_t* __restrict__ stream, int32_t* __restrict__ head,
                                                                  const uint8_t* __restrict__ mask, int n_frames, int S, int N,
                                                                  int C, int HW) {
  const int n = blockIdx.x;
  if (mask && !mask[n]) return;                                   // block-uniform
  const int h0 = head[n];
  const int c0 = C - n_frames;
  for (int k = 0; k < n_frames; ++k) {
    const uint8_t* src = stacks + ((size_t)n * C + c0 + k) * HW;
    uint8_t* dst = stream + ((size_t)((h0 + 1 + k) % S) * N + n) * HW;
    for (int p = threadIdx.x * 16; p < HW; p += FR_THREADS * 16)
      *reinterpret_cast<uint4*>(dst + p) = *reinterpret_cast<const uint4*>(src + p);
  }
  __syncthreads();
  if (threadIdx.x == 0) head[n] = h0 + n_frames;
}

extern "C" int trl_frame_stream_append_u8(const uint8_t* stacks, uint8_t* stream, int32_t* head, const uint8_t* mask,
                                          int n_frames, int S, int N, int C, int HW, void* stream_) {
  TRL_REQUIRE(N >= 0 && C > 0 && S >= C + 1 && HW > 0 && HW % 16 == 0, "bad sizes (HW must be a multiple of 16, S > C)");
  TRL_REQUIRE(n_frames == 1 || n_frames == C, "n_frames must be 1 (newest frame) or C (whole stack)");
  if (N == 0) return TRL_OK;
  TRL_REQUIRE(stacks && stream && head, "null pointer");
  hipLaunchKernelGGL(frame_append_kernel, dim3(N), dim3(FR_THREADS), 0, (hipStream_t)stream_, stacks, stream, head, mask,
                     n_frames, S, N, C, HW);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}

// out[(r * N + n)][c] = stream[(pos[row_idx[r]][n] - C + 1 + shift + c) % S][n]; shift 0 = obs, 1 = next_obs.
// *overrun is set when a requested frame has already been overwritten (position <= head - S).
__global__ __launch_bounds__(FR_THREADS) void frame_gather_kernel(const uint8_t* __restrict__ stream,
                                                                  const int32_t* __restrict__ pos,
                                                                  const int64_t* __restrict__ row_idx, int shift,
                                                                  uint8_t* __restrict__ out, const int32_t* __restrict__ head,
                                                                  int32_t* __restrict__ overrun, int S, int N, int C, int HW) {
  const int n = blockIdx.x, r = blockIdx.y;
  const int p_new = pos[(size_t)row_idx[r] * N + n] + shift;
  if (threadIdx.x == 0 && (p_new - C + 1 <= head[n] - S || p_new > head[n])) atomicOr(overrun, 1);
  for (int c = 0; c < C; ++c) {
    const int sp = p_new - C + 1 + c;
    const uint8_t* src = stream + ((size_t)(((sp % S) + S) % S) * N + n) * HW;
    uint8_t* dst = out + (((size_t)r * N + n) * C + c) * HW;
    for (int p = threadIdx.x * 16; p < HW; p += FR_THREADS * 16)
      *reinterpret_cast<uint4*>(dst + p) = *reinterpret_cast<const uint4*>(src + p);
  }
}

extern "C" int trl_frame_stream_gather_u8(const uint8_t* stream, const int32_t* pos, const int64_t* row_idx, int n_rows,
                                          int shift, uint8_t* out, const int32_t* head, int32_t* overrun, int S, int N,
                                          int C, int HW, void* stream_) {
  TRL_REQUIRE(N >= 0 && n_rows >= 0 && C > 0 && S >= C + 1 && HW > 0 && HW % 16 == 0, "bad sizes");
  TRL_REQUIRE(shift == 0 || shift == 1, "shift is 0 (obs) or 1 (next_obs)");
  if (N == 0 || n_rows == 0) return TRL_OK;
  TRL_REQUIRE(stream && pos && row_idx && out && head && overrun, "null pointer");
  hipLaunchKernelGGL(frame_gather_kernel, dim3(N, n_rows), dim3(FR_THREADS), 0, (hipStream_t)stream_, stream, pos, row_idx,
                     shift, out, head, overrun, S, N, C, HW);
  TRL_LAUNCH_CHECK();
  return TRL_OK;
}
