// Host helper of the reference's exploration-noise stream (torchrl/policies/distribution.py:60-76: one
// `Normal(0, 1).sample()` of (N_total, A) per vector step from the CPU torch generator).  torchrl_amd/collector/noise.py
// derives the generator state at the start of every chunk this process has to produce (trl_mt19937_states_at) -- the
// segments of one block, or this rank's rows of every step's draw when envs are sharded over ranks -- and this library
// fills the chunks with torch's OWN `normal_()` on private generators, several chunks at a time on plain threads: no
// interpreter lock, no Python call per chunk (128 chunks of 12 288 values per rollout at cfg 4; from Python threads the
// per-call overhead serialises them: 4-6 ms per block measured against 0.9 ms for the same values as 8 big segments).
// Built separately from libtrl_hip.so because it links libtorch (ATen's generator and distribution code are what makes
// the values the reference's, bit for bit); plain C ABI, no torch types in the signature.  Optional: without it noise.py
// draws the chunks from Python threads.
#include <ATen/ATen.h>
#include <ATen/CPUGeneratorImpl.h>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <exception>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

static thread_local std::string g_noise_err;

extern "C" const char* trl_noise_last_error(void) { return g_noise_err.c_str(); }
extern "C" int trl_noise_abi_version(void) { return 1; }
// the torch build this helper was compiled against ("<torch.__version__> abi<0|1>", from the build command): the loader
// refuses a helper made for another torch -- its rpath would pull a second libtorch into the process
#ifndef TRL_NOISE_BUILT_FOR
#define TRL_NOISE_BUILT_FOR "unknown"
#endif
extern "C" const char* trl_noise_built_for(void) { return TRL_NOISE_BUILT_FOR; }

// chunk k: generator state image states[k * state_bytes ..) -> out[out_off[k] .. out_off[k] + out_len[k]) standard normals
extern "C" int trl_noise_draw_chunks(const uint8_t* states, int64_t state_bytes, int64_t n_chunks, float* out,
                                     const int64_t* out_off, const int64_t* out_len, int threads) {
  if (!states || !out || !out_off || !out_len || state_bytes <= 0 || n_chunks < 0) {
    g_noise_err = "trl_noise_draw_chunks: bad arguments";
    return -22;
  }
  if (n_chunks == 0) return 0;
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(threads, n_chunks));
  std::atomic<int64_t> next{0};
  std::atomic<bool> failed{false};
  std::string first_error;
  std::mutex err_mutex;
  auto work = [&]() {
    try {
      at::Generator gen = at::detail::createCPUGenerator();
      at::Tensor st = at::empty({state_bytes}, at::kByte);
      for (;;) {
        const int64_t k = next.fetch_add(1);
        if (k >= n_chunks || failed.load()) break;
        std::memcpy(st.data_ptr<uint8_t>(), states + k * state_bytes, (size_t)state_bytes);
        gen.set_state(st);
        at::Tensor o = at::from_blob(out + out_off[k], {out_len[k]}, at::kFloat);
        o.normal_(0.0, 1.0, gen);
      }
    } catch (const std::exception& e) {
      std::lock_guard<std::mutex> lock(err_mutex);
      if (!failed.exchange(true)) first_error = e.what();
    } catch (...) {
      std::lock_guard<std::mutex> lock(err_mutex);
      if (!failed.exchange(true)) first_error = "unknown exception";
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < nt; ++t) pool.emplace_back(work);
  work();
  for (auto& th : pool) th.join();
  if (failed.load()) {
    g_noise_err = "trl_noise_draw_chunks: " + first_error;
    return -1;
  }
  return 0;
}
