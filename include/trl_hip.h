/* trl_hip.h -- C ABI of libtrl_hip.so: the MI355X (gfx950) kernels behind the
 * torchrl collector -> replay_buffer -> algo.update hot path.
 *
 * The reference (RchalYang/torchrl) is 100 % Python and has no FFI; the entry
 * points below are what a maintainer would bind from the reference's own
 * classes (ctypes stubs: INTEGRATION.md).  Each one cites the reference code
 * (file:line under the reference root) whose arithmetic it replaces.
 *
 * Conventions
 *  - every data pointer is a DEVICE pointer into caller-owned, contiguous
 *    memory; nothing is allocated, freed or retained by the library.  The
 *    trl_*_t descriptor structs themselves are HOST memory, read during the
 *    call only;
 *  - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*);
 *  - return 0 on success, a negative TRL_E* code for a rejected argument, or a
 *    positive hipError_t; trl_last_error() gives a thread-local message;
 *  - rollout / replay tensors are time-major fp32: key[row][env][feat]
 *    (reference layout, torchrl/replay_buffers/base.py:22-28, stored fp32
 *    instead of float64); bool keys (terminals, time_limits) are 0.0f / 1.0f;
 *  - an "MLP2" parameter block is ONE flat fp32 buffer
 *      W1[H][D] b1[H] W2[H][H] b2[H] W3[O][H] b3[O] (logstd[O] for a policy)
 *    i.e. nn.Linear's (out, in) layout, layer after layer -- Net/MLPBase of
 *    torchrl/networks/nets.py:13-52, base.py:8-44 (activation after every
 *    hidden layer, linear head).
 */
#ifndef TRL_HIP_H
#define TRL_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define TRL_OK            0
#define TRL_EINVAL       -1   /* bad size / null pointer / misaligned */
#define TRL_EUNSUPPORTED -2   /* shape not instantiated in this build  */

#define TRL_ACT_TANH 0
#define TRL_ACT_RELU 1
#define TRL_ACT_NONE 2

const char* trl_last_error(void);
int trl_abi_version(void);

/* --- K4: GAE / discounted-return reverse scans ---------------------------
 * replaces OnPolicyReplayBufferBase.generalized_advantage_estimation
 * (torchrl/replay_buffers/on_policy.py:16-44) and discount_reward (:46-70).
 * rewards, values, terminals, time_limits, advs, rets: (T, N) fp32;
 * last_value: (N).  time_limits may be NULL when tl_filter == 0.  If
 * last_terminal (N) is given, last_value is masked by (1 - last_terminal) first
 * (OnRLAlgo.process_epoch_samples, torchrl/algo/on_policy/on_rl_algo.py:27). */
int trl_gae_f32(const float* rewards, const float* values, const float* terminals,
                const float* time_limits, const float* last_value, const float* last_terminal,
                float* advs, float* rets, int T, int N,
                float gamma, float tau, int tl_filter, void* stream);
int trl_discount_reward_f32(const float* rewards, const float* values, const float* terminals,
                            const float* time_limits, const float* last_value, const float* last_terminal,
                            float* advs, float* rets, int T, int N,
                            float gamma, int tl_filter, void* stream);

/* --- K5/K6: minibatch / replay gather by time-row index --------------------
 * replaces the fancy-index copies of one_iteration (on_policy.py:72-91) and
 * random_batch (base.py:39-51): dst[i, :] = src[row_idx[i], :], a row being
 * N*F contiguous elements.  row_idx is produced on the host by numpy's legacy
 * global RNG so the index stream is bit-exact with the reference. */
int trl_gather_rows_f32(const float* src, const int64_t* row_idx, float* dst,
                        int n_rows, int64_t row_elems, int64_t src_rows, void* stream);
int trl_gather_rows_u8(const uint8_t* src, const int64_t* row_idx, uint8_t* dst,
                       int n_rows, int64_t row_bytes, int64_t src_rows, void* stream);
/* the same gather for up to 8 keys of one replay sample (random_batch's loop over sample_key, base.py:46-50) in one
 * launch: dst[k][i, :] = src[k][row_idx[i], :], rows of row_bytes[k] bytes; every src[k] has src_rows rows.
 * update_count (nullable): the index row set is then picked on the device -- row_idx is a SLAB {first, sets,
 * idx[sets][n_rows]} (int64), set = (int64)update_count[0] - first; outside [0, sets) nothing is copied.  The `opt_times`
 * samples of one OffRLAlgo.update_per_epoch (off_rl_algo.py:58-66) are drawn up front on the host, in the reference's
 * order, uploaded once, and each update of the captured epoch graph gathers its own. */
int trl_gather_rows_multi(const void* const* src, void* const* dst, const int64_t* row_bytes, int n_keys,
                          const int64_t* row_idx, const double* update_count, int n_rows, int64_t src_rows,
                          void* stream);

/* --- K7: per-minibatch advantage statistics --------------------------------
 * replaces advs.mean()/std()/max()/min() of PPO.update (ppo.py:141-144) for
 * ALL minibatches of an epoch at once.  advs: (T, N); row_idx: (n_mb, rows_mb).
 * raw_out: (n_mb, 4) float64 = {sum, sum of squares, max, -min} -- additive /
 * max-reducible across ranks; the consumer derives mean and the UNBIASED std
 * (ppo.py:147) from them and the global element count. */
int trl_adv_stats_f64(const float* advs, const int64_t* row_idx, int n_mb, int rows_mb,
                      int N, double* raw_out, void* stream);
/* The same as the head of an epoch's update sequence, with what else precedes the minibatch updates riding in the one
 * launch (ppo.py:27-39, utils.py:23-26): a minibatch's rows are cut into up to 8 slices, one workgroup each (40
 * minibatches: 320 workgroups), whose partials the last slice to arrive folds in slice order (deterministic; same
 * quantities as trl_adv_stats_f64, another summation order); `zero_doubles` doubles at `zero` are cleared (the block the
 * updates file their statistics into) and n_copies (<= 4) runs of copy_words[q] 4-byte words are copied from copy_src[q]
 * to copy_dst[q] (target_pf <- pf; the row indices and learning rates of the epoch) -- each optional (0).
 * row_idx and the copy sources may be page-locked HOST memory (the device reads it in place): no copy command in front
 * of the launch then.  workspace: trl_ppo_epoch_prologue_workspace(n_mb) bytes, zeroed ONCE by the caller and then
 * left alone (arrival counters); NULL: one workgroup per minibatch, no workspace. */
int64_t trl_ppo_epoch_prologue_workspace(int n_mb);
int trl_ppo_epoch_prologue_f64(const float* advs, const int64_t* row_idx, int n_mb, int rows_mb, int N, double* raw_out,
                               void* workspace, double* zero, int64_t zero_doubles, int n_copies, void* const* copy_dst,
                               const void* const* copy_src, const int64_t* copy_words, void* stream);

/* --- MLP2 inference -------------------------------------------------------
 * replaces Net.forward (nets.py:49-52) for vf(last_obs) (on_rl_algo.py:25-26)
 * and the policy mean.  x: (M, D) -> out: (M, O). */
int trl_mlp2_forward_f32(const float* params, const float* x, float* out,
                         int M, int D, int H, int O, int act, void* stream);
/* 1 if trl_mlp2_forward_f32 is instantiated for this shape (callers fall back to trl_linear_fwd_f32 per layer) */
int trl_mlp2_forward_supported(int D, int H, int O);

/* --- K1+K2+K3: fused vectorised rollout on the synthetic env ---------------
 * replaces VecOnPolicyCollector.take_actions x n_steps
 * (torchrl/collector/on_policy.py:90-155, base.py:108-122): policy mean ->
 * tanh(mean + std*eps) (continuous_policy.py:92-132, distribution.py:60-76),
 * vf(obs), env step, episode bookkeeping, over-length bootstrap
 * r += gamma*vf(next_obs), terminals = done|surpass, partial reset, and the
 * ring-buffer row write (replay_buffers/base.py:19-37).  See trl_rollout_t. */
typedef struct trl_rollout_t {
  /* networks (MLP2 blocks, D -> H -> H -> {A,1}) */
  const float* pf_params;     /* incl. logstd[A] at the tail */
  const float* vf_params;
  int D, H, A, act;           /* act: TRL_ACT_* */
  int tanh_action;            /* policies' tanh_action flag (continuous_policy.py:92-132) */
  /* synthetic env: obs' = tanh(obs @ env_A + act @ env_B) */
  const float* env_A;         /* (D, D) */
  const float* env_B;         /* (A, D) */
  float reward_scale;
  int horizon;                /* env done = time_limit = (t_env >= horizon) */
  int64_t env_seed_base;      /* env i uses seed env_seed_base + i (vecenv.py:63-65) */
  /* persistent per-env state, all length N */
  float*   cur_obs;           /* (N, D) */
  int32_t* t_env;             /* steps since reset (env side) */
  int32_t* cur_step;          /* collector side counter (base.py:180, 205) */
  int32_t* episode_idx;
  float*   ep_return;         /* running episode return (base.py:181, 216-219) */
  /* noise: host-drawn eps (n_steps, N, A) for reference parity, or NULL ->
   * device Philox keyed by (env seed, noise_step0 + t) */
  const float* noise;
  int64_t noise_step0;
  int deterministic;          /* 1: eps = 0 (greedy eval_act, continuous_policy.py:78-83) */
  /* ring buffer rows [top, top+n_steps) mod rows ; (rows, N, feat) each.
   * All seven may be NULL together: nothing is stored (evaluation rollouts). */
  float *obs, *next_obs, *acts, *values, *rewards, *terminals, *time_limits;
  float *old_logp;            /* optional extra key: log pi(a|s) at collection time */
  int rows, top;
  int N, n_steps;
  int max_episode_frames;
  float discount;
  /* outputs */
  double*  epoch_reward;      /* += sum of raw env rewards (base.py:230) */
  int32_t* ep_count;          /* finished-episode log: count, then entries */
  float*   ep_log;            /* (cap, 3): step, env, return */
  int ep_cap;
  int step0;                  /* value written as `step` for the first step */
  /* running observation normaliser of the env (NormObs, torchrl/env/base_wrapper.py:98-121); norm_state
   * NULL = none.  With it, cur_obs stays the env's RAW state and policy_obs (N, D; in/out) is what the
   * policy sees; stored obs = policy input, stored next_obs = normalised next observation.  When
   * norm_update != 0 the statistics are updated every step from ALL envs: the workgroups rendezvous once
   * per step, so N must not exceed trl_rollout_norm_max_envs(); after any reset the next policy input is
   * the RAW observation of all envs, as the reference's forwarded partial_reset delivers it
   * (normalize_partial_reset != 0: the filtered one instead). */
  double* norm_state;         /* (2D + 1) mean | var | count, see trl_norm_update_filt_f32 */
  float*  policy_obs;
  double* norm_workspace;     /* trl_rollout_norm_workspace(N) doubles, zeroed once by the caller */
  float norm_clip;
  int norm_update;
  int normalize_partial_reset;
  /* 16 bytes (NULL: none) that the launch sets to zero: the {epoch_reward, ep_count} header of the NEXT launch when the
   * caller alternates between two headers -- the memset launch in front of every rollout goes away.  Must not be the
   * header this launch accumulates into. */
  double* clear_header;
  /* (N) floats, NULL: none.  V(next_obs) of the LAST of the n_steps stored steps, from the value pass that follows the
   * rollout in this call (it reads that row anyway): the bootstrap value of on_rl_algo.py:25-26 without a forward launch
   * of its own.  Only with ring tensors. */
  float* boot_values;
  /* publish_words 4-byte words (0: none) copied from publish_src (device) to publish_dst -- page-locked HOST memory,
   * written in place -- by the value pass: the epoch header and the head of the episode log reach the host without a
   * copy command behind the launch.  Only with ring tensors; the words are final when the call's launches have completed. */
  void* publish_dst; const void* publish_src; int64_t publish_words;
  /* noise_flag non-NULL: `noise` is being filled by trl_stage_h2d_f32 on ANOTHER stream; the launch starts its steps
   * only once *noise_flag == noise_stamp (device-scope loads; bounded wait -- on expiry epoch_reward is set to NaN).
   * No stream dependency is needed between the staging launch and this one. */
  const uint32_t* noise_flag; uint32_t noise_stamp;
  /* stage_n != 0: a few EXTRA workgroups of this launch move the NEXT rollout's noise block -- stage_n floats
   * (n % 4 == 0) from page-locked host memory stage_src to the device buffer stage_dst -- while the rollout's own
   * workgroups (half the chip) step the envs: the transfer shares the device with nothing that is sensitive to it.
   * They start once the host has published *stage_ready == stage_job (page-locked word, written after the block is
   * complete; they wait ~30 us for it and otherwise leave without a trace: a host that is not running ahead of the device
   * has not drawn the block yet and stages it itself later), and when the block is in device memory
   * they set stage_state[0] = stage_job (device memory {stamp, arrival counter}, zeroed by the caller; the next launch's
   * noise_flag / noise_stamp) and *stage_ack = stage_job (page-locked: the host learns that the block was staged).
   * Not with a running observation normaliser. */
  const float* stage_src; float* stage_dst; int64_t stage_n;
  const uint32_t* stage_ready; uint32_t stage_job; uint32_t* stage_state; uint32_t* stage_ack;
  /* hipEvent_t or NULL: the VALUE PASS of this call (not the rollout in front of it) waits for this event
   * (hipStreamWaitEvent on `stream`).  The rollout reads only the policy; the value function's parameters may still be
   * in the hands of an update sequence on another stream (the critic's half of PPO.update_per_epoch, ppo.py:93-122, runs
   * beside the actor's and beside the NEXT rollout): the event is that sequence's end. */
  void* value_wait_event;
} trl_rollout_t;
int trl_rollout_synth_f32(const trl_rollout_t* args, void* stream);
/* 1 when trl_rollout_synth_f32 carries networks of this shape (without a running observation normaliser): the
 * benchmark shape D = 17 / A = 6 as a compile-time instantiation, and any 64-wide two-layer policy / value pair with
 * 2..32 inputs and 1..8 actions through the runtime-dims instantiations (17- and 32-feature tiles, missing features
 * and actions masked) -- torchrl/networks/base.py:8-44 and torchrl/collector/on_policy.py:90-155 are shape-generic. */
int trl_rollout_supported(int D, int H, int A, int act);
/* Page-locked host block -> device buffer by a KERNEL (the device reads host memory in place), meant for a stream of
 * its own next to the one that computes: n floats (n % 4 == 0, 16-byte aligned pointers); when every workgroup's part is
 * in device memory, state[0] = stamp is stored with device scope -- what a consumer launched on another stream polls
 * (trl_rollout_t.noise_flag) instead of a stream-to-stream event dependency (which costs sporadic multi-millisecond host
 * stalls on this runtime).  state: 2 x uint32 of device memory {stamp, arrival counter}, zeroed once by the caller. */
int trl_stage_h2d_f32(const float* host_src, float* dev_dst, int64_t n, uint32_t* state, uint32_t stamp, void* stream);
int trl_rollout_norm_workspace(int N);
int trl_rollout_norm_max_envs(int D, int H, int A, int act);

/* synthetic env (re)start: for every env i with mask[i] != 0 (mask NULL = all):
 * episode_idx += 1, obs ~ N(0,1) from the Philox reset stream keyed
 * (env_seed_base + i, episode_idx), env step counter cleared (the collector-side counter and
 * running return only on a full reset, mask == NULL).
 * replaces VecEnv.reset / partial_reset (torchrl/env/vecenv.py:41-51). */
int trl_synth_reset_f32(float* cur_obs, int32_t* t_env, int32_t* cur_step, int32_t* episode_idx,
                        float* ep_return, const uint8_t* mask, int N, int D, int64_t env_seed_base,
                        void* stream);

/* log pi(a|s) of a diagonal Gaussian / TanhNormal given the policy mean:
 * replaces dis.log_prob(actions).sum(-1) of GuassianContPolicyBase.update
 * (torchrl/policies/continuous_policy.py:134-142, distribution.py:33-45).
 * mean, acts: (B, A); logstd: (A); out: (B). */
int trl_gauss_logp_f32(const float* mean, const float* acts, const float* logstd, float* out,
                       int B, int A, int tanh_action, void* stream);

/* --- K8+K9+K10: fused PPO minibatch gradient -------------------------------
 * replaces, for one minibatch, PPO.update's forward/backward work
 * (torchrl/algo/on_policy/ppo.py:41-152): advantage normalisation, vf forward,
 * MSE / clipped value loss, pf forward, TanhNormal log-prob + entropy
 * (distribution.py:33-45, 78-79), ratio / clip / min surrogate, and the
 * backward pass of both MLPs on fp32 MFMA.  Writes per-workgroup partial
 * gradients; trl_ppo_reduce_f32 folds them. */
#define TRL_LOSS_PPO_CLIP 0
#define TRL_LOSS_A2C      1
typedef struct trl_ppo_batch_t {
  const float *obs, *acts, *advs, *rets, *old_values, *old_logp;  /* (rows, N, feat) */
  const int64_t* row_idx;     /* (rows_mb) time rows of this minibatch, or NULL = rows 0.. */
  int rows_mb, N;             /* local samples = rows_mb * N */
  const double* adv_raw;      /* (4) {sum, sumsq, max, -min} of this minibatch (global) */
  double n_global;            /* global sample count (all ranks) */
  const float* pf_params; const float* vf_params;
  int D, H, A, act;
  float clip_para, entropy_coeff;
  int clipped_value_loss, tanh_action;
  int loss_mode;              /* TRL_LOSS_PPO_CLIP: ratio / clip / min surrogate (ppo.py:41-91);
                                 TRL_LOSS_A2C: -mean(log pi * adv) (a2c.py:69-70; old_logp may be NULL) */
  float* partial;             /* (n_wg, P_STRIDE) fp32 workspace */
  double* scal_partial;       /* (n_wg, 8) */
  int n_wg;                   /* workgroups launched (>= 2); rows of partial / scal_partial */
  int n_wg_pf;                /* workgroups [0, n_wg_pf) run the policy, the rest the value net;
                                 0 = even split.  trl_ppo_wg_split() returns the balanced choice.
                                 n_wg_pf = n_wg: ALL workgroups run the policy (vf_params may be NULL);
                                 n_wg_pf = -1: all run the value net (pf_params may be NULL) -- one network
                                 per launch, for trl_ppo_reduce_adam_net_f32 */
} trl_ppo_batch_t;
int trl_ppo_partial_stride(int D, int H, int A);
/* balanced policy / value split of n_wg workgroups for n_tiles = ceil(samples / 16) tiles */
int trl_ppo_wg_split(int D, int H, int A, int n_tiles, int n_wg);
int trl_ppo_minibatch_grad_f32(const trl_ppo_batch_t* args, void* stream);
/* grads: flat [pf grads (P_pf) | vf grads (P_vf)].  info: (24) doubles =
 *  0 sum_j -min(s1,s2) (A2C: sum_j -logp*adv)   1 sum logp   2 sum logp^2   3 max logp   4 -min logp
 *  5 max ratio   6 -min ratio   7 value-loss sum (local samples; divide by the count)
 *  8..11 mean / unbiased std / max / min of the clamped logstd (ppo.py:82-85)
 *  12..15 sum / sum of squares / max / -min of the value prediction (a2c.py:89-92)
 *  16..19 mean / unbiased std over the A dims / max / min of std = exp(clamped logstd) (a2c.py:95-100)
 *  20..23 unused
 * pf_params may be NULL (then 8..11 are left untouched). */
int trl_ppo_reduce_f32(const float* partial, const double* scal_partial, int n_wg, int n_wg_pf,
                       int D, int H, int A, const float* pf_params, float* grads, double* info,
                       void* stream);

/* --- K7/K8 for arbitrary network shapes: the loss half of an update --------
 * replaces everything of PPO.update_actor / update_critic (ppo.py:41-122) and A2C.update (a2c.py:45-106)
 * that sits BETWEEN the networks' forward and backward passes, for networks the fused kernel
 * (trl_ppo_minibatch_grad_f32) is not instantiated for; the layers themselves run on trl_linear_*:
 *   mean (B, A), v (B): outputs of the policy / value networks on this minibatch;  logstd (A): state-independent
 *   acts (B, A), advs / rets / v_old / old_logp (B): the stored rows;  adv_raw: the minibatch's {sum, sumsq, ...}
 *   from trl_adv_stats_f64 and n_global its sample count (advantage normalisation, ppo.py:141-147)
 *   -> d_mean (B, A), d_v (B), d_logstd (A) and info (24 doubles, trl_ppo_reduce_f32's layout; the sums are
 *      over the local samples, divide by the count on the host).
 * Same per-sample arithmetic as the fused kernel.  workspace: trl_ppo_generic_losses_workspace(B, A) doubles. */
int trl_ppo_generic_losses_workspace(int B, int A);
int trl_ppo_generic_losses_f32(const float* mean, const float* logstd, const float* acts, const float* advs,
                               const float* old_logp, const float* v, const float* rets, const float* v_old,
                               const double* adv_raw, double n_global, int B, int A, float clip_para,
                               float entropy_coeff, int clipped_value_loss, int tanh_action, int loss_mode,
                               float* d_mean, float* d_v, float* d_logstd, double* info, double* workspace,
                               void* stream);

/* --- V-MPO: the loss half of VMPO.update (torchrl/algo/on_policy/v_mpo.py:57-181) ----------------------
 * trl_adv_normalize_f32: out = (adv - mean) / (std_unbiased + eps) from trl_adv_stats_f64's {sum, sumsq, ..} (:175-177;
 *   eps = 1e-5 there and in ppo.py / a2c.py, 1e-4 in trpo.py:168).
 * trl_mse_value_loss_f32: d_v = 2 (v - R) / n_global and the loss SUM over the local samples (:136-153).
 * trl_vmpo_losses_f32: on the n samples the host selected (top half by normalised advantage, :64-70):
 *   phi = softmax(adv_n / eta); L_pi = mean(-phi log pi + alpha KL(pi || pi_target)) -> d_mean (n, A), d_logstd (A);
 *   gradients of the dual variables and their Adam(dual_lr, eps 1e-5) step, clamped at 1e-8 (:83-117).
 *   dual_state: 7 floats on the device {eta, alpha, exp_avg x2, exp_avg_sq x2, steps}, initialise {1, 0.1, 0, 0, 0, 0, 0}.
 *   info (12 doubles): 0 policy loss; 1..4 log pi mean / unbiased std / max / min; 5..8 the same of the KL;
 *   9 alpha loss; 10 alpha and 11 eta AFTER the step.  workspace: trl_vmpo_losses_workspace(n, A) doubles. */
int trl_adv_normalize_f32(const float* advs, const double* adv_raw, double n_global, int B, float eps, float* out,
                          void* stream);
int trl_mse_value_loss_f32(const float* v, const float* rets, int B, double n_global, float* d_v, double* loss_sum,
                           void* stream);
int trl_vmpo_losses_workspace(int n, int A);
int trl_vmpo_losses_f32(const float* mean, const float* target_mean, const float* logstd, const float* target_logstd,
                        const float* acts, const float* adv_n, float* dual_state, int n, int A, int tanh_action,
                        float eta_eps, float alpha_eps, float dual_lr, float* d_mean, float* d_logstd, double* info,
                        double* workspace, void* stream);

/* --- TRPO: the element-wise pieces of TRPO.update (torchrl/algo/on_policy/trpo.py:28-226) -----------------
 * trl_trpo_surrogate_f32: L = -mean(p / (p.detach() + 1e-8) * adv_n) - c_ent * mean(ent) (:170-180) on the whole batch:
 *   d_mean (n, A), d_logstd (A); info (5 doubles): 0 L, 1..4 log pi mean / unbiased std / max / min.
 *   workspace: trl_trpo_surrogate_workspace(n, A) doubles.
 * Fisher-vector product F v of the mean KL(pi_theta || pi_theta.detach()) (:62-87) for a diagonal Gaussian policy
 * = backward(forward-mode(v) * exp(-2 logstd) / n) on the network parameters and 2 v on each logstd:
 *   trl_jvp_gate_f32: one layer of the forward-mode pass, out = act'(h) * (a + b) (b, h nullable; a = x W_v^T + b_v
 *   and b = dx W^T come from trl_linear_fwd_f32);  trl_fisher_scale_f32: out = d_mu * exp(-2 logstd) / n.
 * trl_ratio_loss_f32: the line search's -mean(exp(log pi_new - log pi_old) * adv_n) (:110-128), one double. */
int trl_trpo_surrogate_workspace(int n, int A);
int trl_trpo_surrogate_f32(const float* mean, const float* logstd, const float* acts, const float* adv_n, int n, int A,
                           int tanh_action, float entropy_coeff, float* d_mean, float* d_logstd, double* info,
                           double* workspace, void* stream);
int trl_jvp_gate_f32(const float* a, const float* b, const float* h, int act, int64_t n, float* out, void* stream);
int trl_fisher_scale_f32(const float* d_mu, const float* logstd, int n, int A, float* out, void* stream);
int trl_ratio_loss_f32(const float* logp_new, const float* logp_old, const float* adv_n, int n, double* out, void* stream);

/* --- K11: global-norm clip + Adam ------------------------------------------
 * replaces clip_grad_norm_(params, max_norm) + Adam(eps).step()
 * (ppo.py:72-74, 117-119; a2c.py:29-39) for up to 4 parameter groups laid out
 * back to back in params/grads/exp_avg/exp_avg_sq.  step_count is the
 * post-increment Adam step (1 on the first call).  norms_out: (n_groups). */
typedef struct trl_adam_t {
  float* params; const float* grads; float* exp_avg; float* exp_avg_sq;
  int n_groups;               /* <= 4, each clipped by its own global norm */
  int group_sizes[4];
  float group_lr[4];
  float max_norm;             /* <= 0: no clipping */
  float beta1, beta2, eps;
  int step_count;
  float grad_scale;           /* grads are multiplied by this first (1/world_size) */
  float* norms_out;           /* (n_groups) pre-clip global norms */
  int device_state;           /* trl_ppo_reduce_adam_f32 only: 1 = take the step count and the two learning
                                 rates from its workspace header instead of step_count / group_lr
                                 (ws[1]: steps taken so far, uint32; ws[2], ws[3]: lr; ws[4..7]: two doubles
                                 beta1^steps, beta2^steps -- initialise to 1.0; the kernel advances all of them),
                                 so that no launch argument changes between replays of a captured graph */
  double* step_state;         /* trl_clip_adam_f32 only, nullable: 4 doubles on the device {steps taken so far,
                                 beta1^steps, beta2^steps, 0} initialised to {0, 1, 1, 0}.  When set, step_count is
                                 ignored, the step uses steps + 1 and the state is advanced behind the update --
                                 by the last block that has read it (parameter blocks of up to 32 768 elements; the
                                 4th double is the block counter, zero between launches) or by a one-thread kernel
                                 (same purpose as device_state: a captured graph of a whole update can be replayed) */
  const float* device_lr;     /* trl_clip_adam_f32 only, nullable: n_groups learning rates on the device, used instead
                                 of group_lr (a linear schedule then changes no launch argument either) */
} trl_adam_t;
int trl_clip_adam_f32(const trl_adam_t* args, void* stream);
/* the same followed by the Polyak step target <- (1 - tau) target + tau source of n floats (rl_algo.py:169-176,
 * utils.py:16-20) as two launches: where the device-resident step state would need the one-thread tick launch, the Polyak
 * kernel advances it.  ring (nullable): the Polyak launch -- the last one of an update -- also archives the update's
 * statistics block `raw` (raw_bytes, a multiple of 4) into row (optimiser steps taken before this update) mod slots of
 * `ring`; args->step_state is then required */
int trl_clip_adam_polyak_f32(const trl_adam_t* args, float* target, const float* source, int64_t n, float tau,
                             const void* raw, int raw_bytes, void* ring, int slots, void* stream);
/* The whole tail of an off-policy update as ONE launch (twin_sac_q.py:162-220, algo/utils.py:16-20): the fold of the split
 * weight-gradient partials of every layer (entry k: splits[k] slices of n[k] floats at part[k]; the entries cover
 * args->grads -- which receives the folded gradient -- exactly, in order; trl_fold_partials_multi_f32's summation order),
 * clip_grad_norm_ per group, the Adam steps (args->step_state required: the launch advances it), the Polyak step
 * target[i] <- (1 - tau) target[i] + tau params[target_off + i] for i < target_n, and (ring non-NULL) the filing of the
 * statistics block as trl_clip_adam_polyak_f32 does it.  One grid of <= 256 co-resident workgroups with an in-kernel
 * rendezvous for the norms; workspace: trl_fold_clip_adam_polyak_workspace() bytes, zeroed once (word 0 is set if the
 * rendezvous ever timed out: the parameters are then left untouched).  One process only (the gradient SUM over ranks
 * would sit between the fold and the clip). */
int trl_fold_clip_adam_polyak_workspace(void);
int trl_fold_clip_adam_polyak_f32(int count, const float* const* part, const int* n, const int* splits,
                                  const trl_adam_t* args, float* target, int64_t target_off, int64_t target_n, float tau,
                                  const void* raw, int raw_bytes, void* ring, int slots, void* workspace, void* stream);
/* Single-process fast path: trl_ppo_reduce_f32 + trl_clip_adam_f32 in one launch (the block that
 * finishes the reduction last takes the optimiser step; fixed summation orders, deterministic).
 * adam->grads must equal `grads`, the two groups must be [policy | value]; logstd statistics are read
 * from adam->params.  workspace: trl_ppo_reduce_adam_workspace(D, H, A) floats, zeroed once by the
 * caller and then owned by this entry point.  Not usable when gradients are all-reduced between
 * the two steps (world size > 1): call the two separate entry points there. */
int trl_ppo_reduce_adam_workspace(int D, int H, int A);
int trl_ppo_reduce_adam_f32(const float* partial, const double* scal_partial, int n_wg, int n_wg_pf,
                            int D, int H, int A, float* grads, double* info,
                            const trl_adam_t* adam, float* workspace, void* stream);

/* The whole minibatch step of ONE process as ONE launch: trl_ppo_minibatch_grad_f32 followed, inside the same launch, by
 * what trl_ppo_reduce_adam_f32 does (PPO.update's backward, clip_grad_norm_ and optimizer.step of both networks,
 * torchrl/algo/on_policy/ppo.py:67-75, 113-122) -- the workgroups of the gradient grid meet through device-scope flags,
 * fold the partial rows in trl_ppo_reduce_f32's order and step their own 64 parameters; results are bit-identical to the
 * two-launch sequence.  args->n_wg must not exceed trl_ppo_step_max_workgroups() (every workgroup has to be resident at
 * once: TRL_EUNSUPPORTED otherwise -- use the two launches), adam->params / + P_pf must be args->pf_params / vf_params,
 * adam->grads == grads.  workspace: trl_ppo_step_workspace(D, H, A) floats, zeroed once; it begins with
 * trl_ppo_reduce_adam_f32's workspace (same Adam header), so both routes may be used on it in turn.  A rendezvous that does
 * not complete within ~2 s sets workspace word 0 and info[23] and leaves the parameters untouched. */
int trl_ppo_step_workspace(int D, int H, int A);
int trl_ppo_step_max_workgroups(void);
int trl_ppo_minibatch_step_f32(const trl_ppo_batch_t* args, float* grads, double* info, const trl_adam_t* adam,
                               float* workspace, void* stream);

/* One network's half of trl_ppo_reduce_adam_f32: the n_wg rows of `partial` / `scal_partial` come from a single-network
 * gradient launch (trl_ppo_batch_t.n_wg_pf = n_wg: net 0, the policy; n_wg_pf = -1: net 1, the value function); folds them,
 * clips that group by its own norm and takes its Adam step (adam: the two-group descriptor; only group `net` is stepped,
 * norms_out[net] written).  PPO.update_critic and update_actor (ppo.py:93-122, 41-91) touch disjoint networks, optimisers
 * and statistics, so the two halves are independent launch sequences -- the next rollout needs only the policy's.
 * workspace: trl_ppo_reduce_adam_workspace floats PER NETWORK (each keeps its own Adam header). */
int trl_ppo_reduce_adam_net_f32(const float* partial, const double* scal_partial, int n_wg, int net,
                                int D, int H, int A, float* grads, double* info,
                                const trl_adam_t* adam, float* workspace, void* stream);

typedef struct trl_comm trl_comm_t;   /* opaque communicator, see the collectives section below */
/* --- C1 / C2 / C3: collectives of the multi-GPU path ---------------
 * The reference has no distributed backend; with envs sharded by index over one process per GPU
 * these are the calls its update loop makes between backward and clip_grad_norm_
 * (torchrl/algo/on_policy/ppo.py:72-74, 117-119 -- the clip sees the whole-minibatch gradient; SAC:
 * twin_sac_q.py:166-185) and around the advantage statistics (ppo.py:141-147).
 * trl_comm_t is the one opaque object the library owns.  Two transports:
 *   RCCL       ncclAllReduce on the caller's stream (loaded with dlopen; bandwidth-class messages);
 *   peer       8-byte {value, epoch} granules pushed into peer-mapped (hipIpc / xGMI) uncached buffers and summed
 *              in rank order by every rank: one kernel, no separate barrier, identical results on all ranks,
 *              capturable into a HIP graph like any other launch (latency-class messages).
 * Set-up (host, once): rank 0 calls trl_comm_get_unique_id and the caller's rendezvous distributes the
 * trl_comm_unique_id_bytes() bytes; every rank calls trl_comm_init (unique_id NULL = no RCCL communicator),
 * trl_comm_peer_export, all-gathers the trl_comm_peer_handle_bytes()-byte handles and calls trl_comm_peer_open.
 * All ranks must issue the same sequence of collective calls.  A peer wait that times out (a rank is missing)
 * does not hang the GPU: it sets a flag that trl_comm_error() returns (and clears). */
int trl_comm_unique_id_bytes(void);
int trl_comm_peer_handle_bytes(void);
int trl_comm_max_ranks(void);
int trl_comm_get_unique_id(void* id_out);
int trl_comm_init(trl_comm_t** comm, int rank, int world, const void* unique_id);
int trl_comm_peer_export(trl_comm_t* comm, void* handle_out);
int trl_comm_peer_open(trl_comm_t* comm, const void* handles);
int trl_comm_peer_ready(const trl_comm_t* comm);
/* Resident footprint of the launches that wait inside a kernel for other ranks' data (trl_ppo_reduce_adam_xrank_f32):
 * at most `blocks` blocks of 8 waves stay on the device while a rank's gradient is outstanding; 0 (default) = the kernel's
 * own grid (one block per 64 parameters).  Only ranks SHARING a device need it: the waiting launches of all of them must
 * leave CUs for the gradient kernels they are waiting for (blocks <= CUs / (2 x ranks per device)); results do not depend
 * on it.  trl_comm_peer_buffer_kind: 1 = the peer buffer is uncached device memory (hipDeviceMallocUncached), 0 = that
 * allocation failed and plain hipMalloc memory is used (reported on stderr too), -1 = no peer buffer yet. */
int trl_comm_set_wait_footprint(trl_comm_t* comm, int blocks);
int trl_comm_peer_buffer_kind(const trl_comm_t* comm);
int trl_comm_peer_enable(trl_comm_t* comm, int on);   /* 0 after a failed self-check: everything takes the RCCL route */
int trl_comm_has_rccl(const trl_comm_t* comm);
int trl_comm_error(trl_comm_t* comm);
/* trl_comm_error without waiting for the device (a 4-byte read on a stream of the communicator's own): for a once-per-
 * iteration check by a host that runs ahead of the device; a time-out raised by work still in flight is seen by the next call. */
int trl_comm_error_peek(trl_comm_t* comm);
/* what the first timed-out peer wait was waiting for: out[4] = {region (1 gradient, 2 statistics; 0 none), slot = the rank
 * whose contribution was missing, epoch waited for, epoch tag found}; clears the record (diagnostics of a failed run). */
int trl_comm_error_detail(trl_comm_t* comm, int32_t* out);
/* pre-flight of a one-node multi-GPU run (host only, no communicator): out[3] = {peer access possible from dev_a to dev_b,
 * link type (2 PCIe, 4 xGMI, -1 unknown), hops} -- what the peer transport's posted writes and RCCL's rings will run over. */
int trl_comm_link_info(int dev_a, int dev_b, int32_t* out);
int trl_comm_destroy(trl_comm_t* comm);
/* buf <- SUM over ranks, in place, n floats (C1).  Peer transport up to 12 288 floats, RCCL beyond. */
int trl_allreduce_sum_f32(float* buf, int64_t n, trl_comm_t* comm, void* stream);
/* buf <- reduction over ranks, in place, n doubles (C2 / C3): element i is MAXed when bit (i % period) of
 * max_mask is set, SUMmed otherwise (the {sum, sum of squares, max, -min} rows of trl_adv_stats_f64 are
 * period 4, max_mask 0xC).  Peer transport (n <= 2048); a pure SUM falls back to RCCL. */
int trl_allreduce_f64(double* buf, int64_t n, int period, uint64_t max_mask, trl_comm_t* comm, void* stream);
/* trl_ppo_reduce_adam_f32 with the gradient SUM over ranks between the fold and the clip, in the same launch:
 * per-sample gradients already carry 1 / n_global, every rank ends with bit-identical parameters.  The step
 * count / learning rates come from the workspace header (adam->device_state = 1). */
int trl_ppo_reduce_adam_xrank_f32(const float* partial, const double* scal_partial, int n_wg, int n_wg_pf,
                                  int D, int H, int A, float* grads, double* info,
                                  const trl_adam_t* adam, float* workspace, trl_comm_t* comm, void* stream);
/* trl_ppo_reduce_adam_net_f32 with the gradient SUM over ranks inside: one network's half of the step for a rank that
 * runs the critic's and the actor's updates as two launch sequences (ppo.py:93-122 / 41-91) while its env shards sit on
 * several ranks.  Each sequence counts its own exchanges and owns its network's granules of the gradient region, so the
 * two never meet; all ranks must run the same route (joint launches and single-network launches may alternate between
 * runs, not within one).  A bounded wait footprint (trl_comm_set_wait_footprint) is split between the two sequences. */
int trl_ppo_reduce_adam_xrank_net_f32(const float* partial, const double* scal_partial, int n_wg, int net,
                                      int D, int H, int A, float* grads, double* info,
                                      const trl_adam_t* adam, float* workspace, trl_comm_t* comm, void* stream);

/* --- K10 (generic): dense layers of any shape on fp32 MFMA -----------------
 * replaces nn.Linear + activation forward/backward (torchrl/networks/base.py:30-44,
 * nets.py:34-52) for networks the fused PPO kernels are not instantiated for
 * (256-wide SAC nets, Q nets on [obs, act], FC heads).  Row-major, nn.Linear layout:
 *   fwd          y[M,N]  = act(x[M,K] . w[N,K]^T + bias[N])       (bias may be NULL)
 *   bwd_input    dx[M,K] = (dy * act'(y_gate))[M,N] . w[N,K]      (y_gate NULL: no gating)
 *   bwd_weight   dw[N,K] = (dy * act'(y_gate))^T . x ;  db[N] = column sums   (db may be NULL)
 * act' is expressed through the layer OUTPUT y_gate.  bwd_weight needs a workspace of
 * trl_linear_bwd_weight_workspace(M, K, N) floats (split partials, folded in fixed order). */
int trl_linear_fwd_f32(const float* x, const float* w, const float* bias, float* y,
                       int M, int K, int N, int act, void* stream);
/* Same result with the reduction split over up to 8 workgroup slices when the layer has few rows and a long
 * reduction (the conv nets' first FC layer, 512 x 3136 -> 512: 64 output tiles for 256 CUs); partial products
 * go to `workspace` (trl_linear_fwd_workspace(M, K, N) floats, 0 = the layer is not split and workspace may
 * be NULL) and are folded in fixed order together with bias and activation. */
int trl_linear_fwd_workspace(int M, int K, int N);
/* (…_splitk_group_f32: G same-shaped layers, one launch of split GEMMs + one fold; workspace = G x that many floats) */
int trl_linear_fwd_splitk_group_f32(int G, const float* const* x, const float* const* w, const float* const* bias,
                                    float* const* y, int M, int K, int N, int act, float* workspace, void* stream);
int trl_linear_fwd_splitk_f32(const float* x, const float* w, const float* bias, float* y,
                              int M, int K, int N, int act, float* workspace, void* stream);
int trl_linear_bwd_input_f32(const float* dy, const float* y_gate, int gate_act, const float* w,
                             float* dx, int M, int K, int N, void* stream);
/* the same for few output tiles behind a long reduction (few rows, a wide layer: QR-DQN's 1200-wide head): the reduction over
 * the N outputs is split over workgroup slices, partial dX in `workspace` (trl_linear_bwd_input_workspace floats; 0 = the
 * shape does not split and the call is trl_linear_bwd_input_f32), fixed-order fold -- deterministic */
int trl_linear_bwd_input_workspace(int M, int K, int N);
int trl_linear_bwd_input_splitk_f32(const float* dy, const float* y_gate, int gate_act, const float* w, float* dx,
                                    float* workspace, int M, int K, int N, void* stream);
int trl_linear_bwd_weight_workspace(int M, int K, int N);
int trl_linear_bwd_weight_f32(const float* dy, const float* y_gate, int gate_act, const float* x,
                              float* dw, float* db, float* workspace, int M, int K, int N,
                              void* stream);
/* Grouped forms: G <= 8 independent layers of IDENTICAL shape in one launch (the twin critics qf1 / qf2 of
 * twin_sac_q.py:121-150 and td3.py:96-110, their target copies, one network applied to several inputs).
 * x / w / bias / y ... are HOST arrays of G device pointers; bias (the array, or all of its entries) may be
 * NULL; either every problem of a group is gated / wants db or none.  Results are identical to G separate
 * calls (same kernel, same summation order).  bwd_weight_group needs G * trl_linear_bwd_weight_workspace floats. */
int trl_linear_fwd_group_f32(int G, const float* const* x, const float* const* w, const float* const* bias,
                             float* const* y, int M, int K, int N, int act, void* stream);
int trl_linear_bwd_input_group_f32(int G, const float* const* dy, const float* const* y_gate, int gate_act,
                                   const float* const* w, float* const* dx, int M, int K, int N, void* stream);
int trl_linear_bwd_weight_group_f32(int G, const float* const* dy, const float* const* y_gate, int gate_act,
                                    const float* const* x, float* const* dw, float* const* db, float* workspace,
                                    int M, int K, int N, void* stream);

/* Deferred folds: the GEMM half of trl_linear_bwd_weight_group_f32 only.  Problem i leaves
 * S = trl_linear_bwd_weight_splits(M, K, N) partials of dW at workspace + i * S * (N*K + N), laid out [S][N*K], followed
 * (want_db != 0) by S partials of db, [S][N].  trl_fold_partials_multi_f32 then folds up to 32 such (partials -> out)
 * pairs of n[k] floats in splits[k] partials in ONE launch -- every layer of a backward pass at once -- in the summation
 * order of the folding entry points. */
int trl_linear_bwd_weight_splits(int M, int K, int N);
int trl_linear_bwd_weight_partials_group_f32(int G, const float* const* dy, const float* const* y_gate, int gate_act,
                                             const float* const* x, int want_db, float* workspace, int M, int K,
                                             int N, void* stream);
int trl_fold_partials_multi_f32(int count, const float* const* part, float* const* out, const int* n,
                                const int* splits, void* stream);
/* Fold scope: between begin and end (host state of the calling thread) every single-problem weight-gradient fold that
 * trl_linear_bwd_weight_f32 / trl_conv_bwd_weight_{nhwc,u8}_f32 would launch is recorded instead, and `end` runs them as ONE
 * launch -- same arithmetic and summation order (bit-identical gradients), one dependent launch instead of one per layer
 * (the conv trunk of dqn_pong.json: three folds of ~5 us).  Up to 8 folds per scope (more are launched as they come); the
 * caller gives each layer its own workspace region and reads no gradient before `end`.  Replaces nothing in the reference:
 * it is the launch-count side of autograd's per-layer weight gradients (torchrl/networks/base.py:59-107 backward). */
int trl_fold_scope_begin(void);
int trl_fold_scope_end(void* stream);
/* The weight gradients of up to 12 layers of DIFFERENT widths (K[i] inputs, N[i] outputs, the same batch M) as ONE
 * launch of split GEMMs: problem i leaves S_i = trl_linear_bwd_weight_multi_splits(M, K[i], N[i]) partials of dW_i at
 * workspace[i] ([S_i][N_i * K_i]) followed (want_db) by S_i partials of db_i ([S_i][N_i]); y_gate[i] may be NULL (no
 * activation behind that layer).  Nothing on a backward pass waits for a weight gradient before the optimiser step, so
 * they need not be one dependent launch per layer. */
int trl_linear_bwd_weight_multi_splits(int M, int K, int N);
int trl_linear_bwd_weight_partials_multi_f32(int G, const float* const* dy, const float* const* y_gate, int gate_act,
                                             const float* const* x, const int* K, const int* N, int want_db,
                                             float* const* workspace, int M, void* stream);

/* Forward of G MLPs D -> 256 -> 256 -> O (D <= 32, O <= 16: the policy / Q networks of SAC, DDPG, TD3;
 * networks/base.py:30-44, nets.py:34-68) in ONE launch: hidden activations stay in LDS, W2 is walked in double-buffered
 * k panels.  h1[k] / h2[k] (M, 256) receive the hidden activations of the networks whose backward pass needs them (the
 * array or single entries may be NULL); act after both hidden layers, last_act after the head. */
int trl_mlp3_forward_ok(int D, int H1, int H2, int O);
int trl_mlp3_forward_group_f32(int G, const float* const* x, const float* const* w1, const float* const* b1,
                               const float* const* w2, const float* const* b2, const float* const* w3,
                               const float* const* b3, float* const* h1, float* const* h2, float* const* y, int M,
                               int D, int O, int act, int last_act, void* stream);

/* --- K12 / K13: twin-Q SAC update pieces (torchrl/algo/off_policy/twin_sac_q.py:84-220) ---- */
/* torch.cat([obs, act], -1) of QNet.forward (torchrl/networks/nets.py:61-68) */
int trl_concat2_f32(const float* a, const float* b, float* out, int rows, int fa, int fb, void* stream);
/* pf.explore(x, return_log_probs=True) after the MLP: head (B, 2A) = [mean | log_std] ->
 * action = tanh(mean + std * eps), log_prob (B)   (continuous_policy.py:92-121, 162-170) */
int trl_tanh_gauss_rsample_fwd_f32(const float* head, const float* eps, float* act, float* logp,
                                   int B, int A, int tanh_action, void* stream);
/* its backward plus the std/mean regularisers (twin_sac_q.py:157-160) -> d_head (B, 2A) */
/* d(loss)/d(log_prob) is the same for every row: (*d_logp_ptr) * d_logp_mul (alpha is a device
 * scalar, d_logp_mul = 1 / B); d_logp_ptr NULL means 1 */
int trl_tanh_gauss_rsample_bwd_f32(const float* head, const float* eps, const float* act,
                                   const float* d_act, const float* d_logp_ptr, float d_logp_mul,
                                   float w_std, float w_mean, float* d_head, int B, int A,
                                   int tanh_action, void* stream);
/* the same with d_act = dx1[:, off:off+A] + dx2[:, off:off+A] read in place (rows of ld floats, dx2 may be NULL): the
 * twin critics' input gradients on [obs | new_a] (twin_sac_q.py:152-155) without a slice-and-add launch */
int trl_tanh_gauss_rsample_bwd_cols_f32(const float* head, const float* eps, const float* act, const float* dx1,
                                        const float* dx2, int ld, int off, const float* d_logp_ptr,
                                        float d_logp_mul, float w_std, float w_mean, float* d_head, int B,
                                        int A, int tanh_action, void* stream);
/* The policy gradient from the critics' first hidden layer to the policy head in one streaming launch
 * (twin_sac_q.py:146-160): critic i (n = 1 or 2) contributes dZ_i = dy[i] * act'(y[i]) (both (B, H); y NULL or
 * gate_act NONE: dy is dZ already) times the ACTION columns [off, off + A) of its first-layer weight w[i] (H, ldw),
 *   d_act = sum_i dZ_i w[i][:, off:off+A],   d_head = what trl_tanh_gauss_rsample_bwd_f32 makes of d_act
 * -- the input-gradient GEMM of that layer (whose other columns nobody reads) and the sampler's backward launch in one.
 * H % 4 == 0, H <= 1024, A <= 8 (trl_sac_policy_grad_supported), 16-byte aligned rows.
 * head_dz (nullable): the policy's own head backward rides along (GuassianContPolicy's last nn.Linear, (2A, H) weight head_w):
 *   head_dz = (d_head head_w) * act'(head_h)      (B, H), head_h = that layer's input = the second hidden layer's outputs
 * i.e. the gradient at the policy's second hidden layer already gated for the layer below (the same hidden width H as the
 * critics': nets.py builds both from one hidden_shapes list) -- instead of a 2A-deep input-gradient GEMM launch. */
int trl_sac_policy_grad_supported(int H, int A);
int trl_sac_policy_grad_f32(int n, const float* const* dy, const float* const* y, int gate_act, const float* const* w,
                            int H, int ldw, int off, const float* head, const float* eps, const float* act,
                            const float* d_logp_ptr, float d_logp_mul, float w_std, float w_mean, float* d_head, int B,
                            int A, int tanh_action, const float* head_w, const float* head_h, int head_act, float* head_dz,
                            void* stream);
/* both policy samples of one update and the three critic inputs in ONE launch (twin_sac_q.py:93-106, 125-131,
 * 146-151): (new_a, logp) from head = pf(obs) with eps1, (next_a, next_logp) from head2 = pf(next_obs) with eps2,
 * x_sa = [obs | acts], x_next = [next_obs | next_a], x_new = [obs | new_a]  (each (B, D + A)).
 * step_state (nullable): the two noise draws (distribution.py:67-70) are then made inside the launch: update u -- u =
 * step_state[0], the device-resident count of optimiser steps taken (trl_adam_t.step_state), so the launch can be
 * graph-replayed -- uses trl_philox_normal_f32's (B, A) draws for (seed, 2u + 1) and (seed, 2u + 2); eps1 RECEIVES the
 * first one (the sampler's backward pass reads it), eps2 is not touched.
 * mom_part (nullable): per-wave partial moments of the clamped log_std / log_prob / mean as a by-product (ceil(B/64) rows
 * of 12 doubles), folded by trl_sac_losses_f32. */
int trl_sac_samples_f32(const float* head, const float* head2, float* eps1, const float* eps2,
                        const double* step_state, int64_t seed, const float* obs, const float* acts,
                        const float* next_obs, float* new_a, float* logp, float* next_a, float* next_logp,
                        float* x_sa, float* x_next, float* x_new, int B, int D, int A, int tanh_action,
                        double* mom_part, void* stream);
/* alpha loss + Adam step on log_alpha + alpha = exp(log_alpha) (twin_sac_q.py:111-120).
 * state (4): log_alpha, exp_avg, exp_avg_sq, step; out (2): alpha, alpha_loss */
int trl_sac_alpha_step_f32(const float* logp, int B, float target_entropy, float lr, float beta1,
                           float beta2, float eps, float* state, float* out, void* stream);
/* TD target, twin MSE losses and all loss gradients w.r.t. the Q outputs (twin_sac_q.py:125-155).
 * every tensor (B); alpha: device scalar; sums (4 doubles): qf1 loss sum, qf2 loss sum,
 * sum(alpha logp - min(q1n, q2n)), sum(rewards).  Two optional extras that each save a launch of a single-rank update:
 *  - alpha_state / alpha_out non-NULL: the entropy-temperature step of trl_sac_alpha_step_f32 (twin_sac_q.py:111-120) is
 *    taken first, on `logp`, and its alpha is the one used (the `alpha` argument is ignored and may be NULL);
 *  - mom_part non-NULL: the per-wave partial moments written by trl_sac_samples_f32 (ceil(B/64) rows of 12 doubles)
 *    are folded into mom_out[12] = {mean, unbiased std, max, min} of the clamped log_std, of log_prob and of the mean
 *    (the numbers trl_moments_multi_f64 gives on the policy head, twin_sac_q.py:190-207). */
int trl_sac_losses_f32(const float* q1, const float* q2, const float* tq1, const float* tq2,
                       const float* logp_next, const float* rew, const float* term, const float* q1n,
                       const float* q2n, const float* logp, const float* alpha, float gamma, int B,
                       float* dq1, float* dq2, float* dq1n, float* dq2n, double* sums,
                       float* alpha_state, float* alpha_out, float target_entropy, float lr, float beta1,
                       float beta2, float eps, const double* mom_part, int A, double* mom_out, void* stream);
/* DDPG / TD3 (torchrl/algo/off_policy/ddpg.py:42-110, td3.py:57-154): TD target with Q' = tq1 or
 * min(tq1, tq2) (tq2 NULL: single critic), MSE of one or two critics + output gradients; with qn also the
 * policy loss -mean(Q(s, pi(s))) and dqn = -1/B.  sums (4 doubles): q1 loss sum, q2 loss sum, sum(-qn), sum(r) */
int trl_detac_losses_f32(const float* q1, const float* q2, const float* tq1, const float* tq2,
                         const float* rewards, const float* terminals, const float* qn, float gamma, int B,
                         float* dq1, float* dq2, float* dqn, double* sums, void* stream);
/* out = clamp(act + clamp(sigma * eps, +-noise_clip), lo, hi): FixGuassianContPolicy.explore
 * (continuous_policy.py:67-74) and TD3's target-policy smoothing (td3.py:75-82) */
int trl_noisy_action_f32(const float* act, const float* eps, float sigma, float noise_clip, float lo, float hi,
                         float* out, int64_t n, void* stream);
/* out (rows, A) = x1[:, off:off+A] + x2[:, off:off+A]  (d policy_loss / d action through both Q nets; x2 may be NULL) */
int trl_slice_add_f32(const float* x1, const float* x2, float* out, int rows, int ld, int off, int A,
                      void* stream);
/* d(input) of a layer with ONE output, already gated for the layer below: out[m][f] = dq[m] * w[f] * act'(h[m][f])
 * (h = that layer's activation output, (M, N) like out; N % 4 == 0, 16-byte aligned h / out); G problems per launch.
 * Replaces a K = 1 trl_linear_bwd_input_f32 and the gate operand of the GEMMs that consume its result. */
int trl_outer_gate_group_f32(int G, const float* const* dq, const float* const* w, const float* const* h,
                             float* const* out, int M, int N, int act, void* stream);
/* K13: target <- (1 - tau) target + tau source  (torchrl/algo/utils.py:16-20) */
int trl_polyak_f32(float* target, const float* source, int64_t n, float tau, void* stream);
/* logging: `count` (1..4) statistics in one launch, one entry of every array per statistic: mean / unbiased std / max /
 * min over columns [off, off+width) of rows of `ld` floats, each value clamped to [clamp_lo, clamp_hi] first (the logged
 * log_std is the clamped one).  ring (nullable): the launch also files the update's statistics block `raw` (raw_bytes, a
 * multiple of 8; every out4[k] lies inside it) into slot ((int64)update_count[0] - 1) mod slots of `ring` (slots x
 * raw_bytes): the deferred-update protocol reads a whole epoch's info dicts (off_rl_algo.py:62-64,
 * logger.add_update_info) back in one copy. */
int trl_moments_multi_f64(int count, const float* const* x, const int64_t* n, const int* ld, const int* off,
                          const int* width, const float* clamp_lo, const float* clamp_hi, double* const* out4,
                          const void* raw, int raw_bytes, void* ring, int slots, const double* update_count,
                          void* stream);
/* One VecCollector.take_actions (torchrl/collector/base.py:184-230) on the synthetic vector env in ONE launch, after the
 * policy MLP: action = rsample(head, eps) (as trl_tanh_gauss_rsample_fwd_f32), obs / acts stored, env.step
 * (as trl_synth_env_step_f32: cur_obs advanced in place), next_obs / rewards / terminals / time_limits rows written, the
 * collector's bookkeeping (as trl_collector_bookkeep_f32) and the partial reset of the envs it flags (as
 * trl_synth_reset_f32 with that mask).
 * state NULL: the six pointers are THE rows to write (obs / acts / time_limits may be NULL: evaluation stores nothing),
 * `step` is the logged step; eps NULL: the noise is rows [noise_row0, noise_row0 + N) of trl_philox_normal_f32's (all
 * envs, A) draw for (noise_seed, noise_counter), generated in place (same Philox blocks, no separate launch).
 * state given: every per-step quantity lives on the device, so that the launch (and the policy pass in front of it) is
 * captured once and replayed for every vector step: state = 3 int64 {global step, ring row, first step of the epoch}
 * followed by a zeroed 32-bit block counter at state + 3 (32 bytes in all); the six pointers are the whole ring tensors
 * (n_rows time rows of N envs); the noise is drawn in place (counter = global step, eps must be NULL); the launch stores
 * into row state[1] and advances state[0] and state[1] itself. */
int trl_synth_collect_step_f32(float* cur_obs, const float* head, const float* eps, int64_t noise_seed,
                               int64_t noise_counter, int noise_row0, const float* env_A,
                               const float* env_B, int32_t* t_env, int32_t* cur_step, int32_t* episode_idx,
                               float* ep_return, float reward_scale, int horizon, int max_episode_frames,
                               int64_t env_seed_base, float* obs, float* acts, float* next_obs, float* rewards,
                               float* terminals, float* time_limits, int n_rows, int64_t* state,
                               uint8_t* reset_mask, double* epoch_reward, int32_t* ep_count, float* ep_log,
                               int ep_cap, int step, int N, int D, int A, int tanh_action, void* stream);
/* N(0,1) fill from the Philox4x32-10 stream (device exploration / rsample noise) */
int trl_philox_normal_f32(float* out, int64_t n, int64_t seed, int64_t counter, void* stream);
/* K1 stand-alone: one VecEnv.step of the synthetic env (torchrl/env/vecenv.py:53-61); cur_obs is
 * advanced in place and copied to next_obs; rewards / dones are (N) floats */
int trl_synth_env_step_f32(float* cur_obs, const float* act, const float* env_A, const float* env_B,
                           int32_t* t_env, float reward_scale, int horizon, float* next_obs,
                           float* rewards, float* dones, int N, int D, int A, void* stream);

/* off-policy collector bookkeeping after env.step (torchrl/collector/base.py:205-224): step
 * counters, running returns (logged + cleared on done), reset_mask = done | step >= max frames */
int trl_collector_bookkeep_f32(const float* rewards, const float* dones, int32_t* cur_step,
                               float* ep_return, int max_episode_frames, uint8_t* reset_mask,
                               double* epoch_reward, int32_t* ep_count, float* ep_log, int ep_cap,
                               int step, int N, void* stream);

/* --- K16: conv layers of CNNBase (torchrl/networks/base.py:59-107) as im2col + the GEMM family ----
 * activations channels-last (B, H, W, C); cols[(b,oy,ox)][c*kh*kw + i*kw + j] matches the
 * nn.Conv2d weight viewed as (Cout, Cin*kh*kw).  No padding.  The u8 variant reads NCHW uint8
 * frame stacks and applies x * scale + shift (ScaledFloatFrame, env/atari_wrapper.py:230-240). */
int trl_im2col_f32(const float* in_nhwc, float* cols, int B, int C, int H, int W, int kh, int kw,
                   int sh, int sw, void* stream);
int trl_im2col_u8_nchw(const uint8_t* in_nchw, float* cols, int B, int C, int H, int W, int kh, int kw,
                       int sh, int sw, float scale, float shift, void* stream);
int trl_col2im_f32(const float* dcols, float* dx_nhwc, int B, int C, int H, int W, int kh, int kw,
                   int sh, int sw, void* stream);
/* d(input) of a conv layer on channels-last activations WITHOUT the cols matrix (autograd's conv2d input gradient in
 * the reference, networks/base.py:59-107): dx (B, H, W, Cin) from dy (B, Ho, Wo, Cout) gated by act'(y_gate) (y_gate =
 * the layer's activation output, nullable) and the nn.Conv2d weight (Cout, Cin, kh, kw) as stored.  An implicit
 * transposed convolution per stride-parity class on the MFMA units; trl_conv_bwd_input_nhwc_ok says whether a geometry
 * is covered (Cin a multiple of 16 up to 64, Cout 16 / 32 / 64, stride <= kernel) -- otherwise trl_linear_bwd_input_f32 +
 * col2im. */
int trl_conv_bwd_input_nhwc_ok(int Cin, int Cout, int kh, int kw, int sh, int sw);
int trl_conv_bwd_input_nhwc_workspace(int Cin, int Cout, int kh, int kw);   /* floats: the weights re-ordered per call */
/* x_gate (nullable, laid out like dx) with x_gate_act: the result is multiplied by act'(x_gate) on the way out, i.e. the
 * previous layer receives its dZ instead of its dY and gates nothing itself.
 * prepped != 0: `workspace` already holds this layer's re-ordered weights (trl_conv_bwd_input_nhwc_prep_f32 re-orders the
 * weights of up to 8 layers in ONE launch -- a trunk's backward pass otherwise pays a ~5 us prep launch per layer) */
int trl_conv_bwd_input_nhwc_prep_f32(int n, const float* const* w, float* const* workspace, const int* Cin, const int* Cout,
                                     const int* kh, const int* kw, const int* sh, const int* sw, void* stream);
int trl_conv_bwd_input_nhwc_f32(const float* dy, const float* y_gate, int gate_act, const float* w, float* dx,
                                const float* x_gate, int x_gate_act, float* workspace, int B, int Cin, int H, int W,
                                int kh, int kw, int sh, int sw, int Cout, int prepped, void* stream);
/* out[b][c][p] = in[b][p][c]  (NCHW flatten order in front of the first FC layer, and back) */
int trl_transpose_bpc_f32(const float* in, float* out, int B, int P, int C, void* stream);
/* the same with out[e] *= act'(y_gate[..]): d(features) -> the last conv layer's dZ.  gate_like_in == 0: y_gate is laid
 * out like out; != 0: like in (the last conv layer's output kept in the flattened order, trl_conv_fwd_nhwc_f32 out_chw) */
int trl_transpose_bpc_gate_f32(const float* in, const float* y_gate, int gate_act, int gate_like_in, float* out, int B,
                               int P, int C, void* stream);

/* --- K16b: first conv layer straight from uint8 frames (implicit GEMM) -------
 * replaces nn.Conv2d + activation of CNNBase's first layer (torchrl/networks/base.py:59-107) applied to
 * ScaledFloatFrame(frames) (torchrl/env/atari_wrapper.py:230-240) WITHOUT materialising the im2col matrix:
 *   y[(b, oy, ox)][co] = act( sum_{c,i,j} (frames[b][c][oy*sh+i][ox*sw+j] * scale + shift) * w[co][c][i][j] + bias[co] )
 * frames: (B, C, H, W) uint8 NCHW; w: nn.Conv2d weight (Cout, C, kh, kw) as stored; y: (B*Ho*Wo, Cout) = NHWC.
 * Requires kw, sw and W to be multiples of 4 (each 4 consecutive taps are one aligned dword); other geometries
 * return TRL_EINVAL and go through trl_im2col_u8_nchw + trl_linear_*.
 * trl_conv_bwd_weight_u8_f32: dw (Cout, C*kh*kw), db (Cout) (nullable) from dy (B*Ho*Wo, Cout), gated by
 * act'(y_gate) as in trl_linear_bwd_weight_f32; workspace: trl_conv_bwd_weight_workspace(...) floats.
 * riders (nullable): small weight re-orderings that ride on the forward launch as extra workgroups -- re-made from the live
 * weights by every forward pass, at no launch of their own:
 *   perm jobs  the weight (perm_cout x perm_c x perm_khw, nn.Conv2d's layout) of a LATER conv layer copied to perm_dst in the
 *              (i, j, c) reduction order of trl_conv_fwd_nhwc_f32, which then takes it with w_perm != 0 (dense 16-byte weight
 *              loads instead of strided 4-byte ones);
 *   dx jobs    what trl_conv_bwd_input_nhwc_prep_f32 makes for a later layer's input gradient (dx_ws:
 *              trl_conv_bwd_input_nhwc_workspace floats), for a backward pass that follows with prepped != 0. */
typedef struct trl_conv_riders_t {
  int n_perm;                 /* <= 4 */
  const float* perm_src[4]; float* perm_dst[4]; int perm_cout[4], perm_c[4], perm_khw[4];
  int n_dx;                   /* <= 4 */
  const float* dx_w[4]; float* dx_ws[4]; int dx_cin[4], dx_cout[4], dx_kh[4], dx_kw[4], dx_sh[4], dx_sw[4];
} trl_conv_riders_t;
int trl_conv_fwd_u8_f32(const uint8_t* frames, const float* w, const float* bias, float* y, int B, int C, int H,
                        int W, int kh, int kw, int sh, int sw, float scale, float shift, int Cout, int act,
                        const trl_conv_riders_t* riders, void* stream);
/* The same layer of TWO networks of one architecture on two frame batches of one shape -- DQN's online net on obs and its
 * target net on next_obs (torchrl/algo/off_policy/dqn.py:38-52) -- as ONE launch (narrow first layers: one ragged last
 * round of workgroups instead of two and one launch boundary less; wide ones: two launches).  `riders` may carry both
 * networks' jobs (at most 4 of either kind together). */
int trl_conv_fwd_u8_pair_f32(const uint8_t* frames_a, const float* w_a, const float* bias_a, float* y_a,
                             const uint8_t* frames_b, const float* w_b, const float* bias_b, float* y_b, int B, int C, int H,
                             int W, int kh, int kw, int sh, int sw, float scale, float shift, int Cout, int act,
                             const trl_conv_riders_t* riders, void* stream);
int trl_conv_bwd_weight_workspace(int B, int C, int H, int W, int kh, int kw, int sh, int sw, int Cout);
/* The later conv layers, same idea on fp32 channels-last activations x (B, H, W, C), C % 4 == 0: the reduction
 * runs in (i, j, c) order so that a window row is one contiguous run of kw*C floats; w is still the nn.Conv2d
 * weight (Cout, C, kh, kw) as stored and dw comes back in that layout.  y: (B*Ho*Wo, Cout) = NHWC, or with
 * out_chw != 0 (B, Cout, Ho*Wo): what nn.Flatten hands the FC layers (networks/base.py:100-107), written by the
 * layer's own epilogue instead of a transposing launch.
 * Workspace of the weight gradient: trl_conv_bwd_weight_workspace. */
int trl_conv_fwd_nhwc_f32(const float* x, const float* w, const float* bias, float* y, int B, int C, int H, int W,
                          int kh, int kw, int sh, int sw, int Cout, int act, int out_chw, int w_perm, void* stream);
/* G conv layers of one geometry (different inputs / weights / outputs) in one launch: the online and the target
 * network of a DQN update (dqn.py:47-52) run the same trunk on obs and next_obs */
int trl_conv_fwd_nhwc_group_f32(int G, const float* const* x, const float* const* w, const float* const* bias,
                                float* const* y, int B, int C, int H, int W, int kh, int kw, int sh, int sw, int Cout,
                                int act, int out_chw, int w_perm, void* stream);
int trl_conv_bwd_weight_nhwc_f32(const float* dy, const float* y_gate, int gate_act, const float* x, float* dw,
                                 float* db, float* workspace, int B, int C, int H, int W, int kh, int kw, int sh,
                                 int sw, int Cout, void* stream);
int trl_conv_bwd_weight_u8_f32(const float* dy, const float* y_gate, int gate_act, const uint8_t* frames,
                               float* dw, float* db, float* workspace, int B, int C, int H, int W, int kh,
                               int kw, int sh, int sw, float scale, float shift, int Cout, void* stream);

/* --- K14: DQN TD loss (torchrl/algo/off_policy/dqn.py:53-60): q, q_next (B, A); dq (B, A); sums (3 doubles):
 * squared-error sum, q_s_a sum, reward sum.  The actions come as int64 (`acts`) or as the floats the replay buffer stores
 * (`acts_f`; the reference's `actions.long()` cast, dqn.py:47, then happens at the read) -- exactly one of the two is
 * non-NULL; a stored value outside [0, A) (or NaN) is clamped instead of indexing out of bounds.  ring (slots x 3
 * doubles, nullable): the three sums are also filed into row ((int64)update_count[0] mod slots) -- the count of updates
 * finished before this one -- so that a captured update needs no copy command for its statistics. */
int trl_dqn_td_loss_f32(const float* q, const int64_t* acts, const float* acts_f, const float* q_next,
                        const float* rewards, const float* terminals, float gamma, int B, int A, float* dq,
                        double* sums, double* ring, int slots, const double* update_count, void* stream);
/* --- K14b: the linear head of a DQN update in one launch (dqn.py:47-60 and nn.Linear's autograd, nets.py:34-52):
 * from the last hidden activations h (online net on obs) and h_next (target net on next_obs), both (B, H):
 *   q = h w^T + bias, q_next = h_next w_t^T + bias_t   ((B, A); written to q_out / qn_out when non-NULL),
 *   K14's loss, sums and ring row,
 *   dh (B, H) = dq w,  dw (A, H) = dq^T h,  db (A) = column sums of dq      (dq as K14 defines it)
 * -- what trl_linear_fwd_*, trl_dqn_td_loss_f32, trl_linear_bwd_weight_f32 and trl_linear_bwd_input_f32 compute as
 * seven launches.  Deterministic (fixed sample -> wave map, fixed fold order).  Shapes: trl_dqn_head_supported
 * (H % 4 == 0, H <= 1024, A <= 8); workspace: trl_dqn_head_workspace bytes, 16-byte aligned, ZEROED ONCE before the
 * first call and then left alone (it holds the launch-to-launch arrival counter of the kernel's internal rendezvous);
 * calls sharing a workspace must be stream-ordered. */
int trl_dqn_head_supported(int H, int A);
int64_t trl_dqn_head_workspace(int H, int A);
int trl_dqn_head_f32(const float* h, const float* h_next, const float* w, const float* bias, const float* w_t,
                     const float* bias_t, const int64_t* acts, const float* acts_f, const float* rewards,
                     const float* terminals, float gamma, int B, int H, int A, float* dh, float* dw, float* db,
                     float* q_out, float* qn_out, double* sums, double* ring, int slots, const double* update_count,
                     void* workspace, void* stream);
/* --- K15: QR-DQN quantile-Huber loss + output gradient (qrdqn.py:39-60, algo/utils.py:5-13):
 * q, q_next, dq (B, A*Q); workspace 2B doubles; acts / acts_f / sums / ring as above (loss sum is over B*Q*Q terms) */
int trl_quantile_huber_f32(const float* q, const int64_t* acts, const float* acts_f, const float* q_next,
                           const float* rewards, const float* terminals, float gamma, int B, int A, int Q,
                           float* dq, double* workspace, double* sums, double* ring, int slots,
                           const double* update_count, void* stream);
/* --- K17: greedy / epsilon-greedy action (torchrl/policies/discrete_policies.py:40-67, 86-89):
 * argmax_a of Q (Q == 1) or of the mean over Q quantiles; where u[n] < epsilon -> rand_act[n] */
int trl_eps_greedy_i64(const float* q, int N, int A, int Q, const float* u, const int64_t* rand_act,
                       float epsilon, int64_t* action, int64_t* ring_row, int n_rows, void* stream);
/* (ring_row, nullable: ring_row[0] = (ring_row[0] + 1) % n_rows rides along -- trl_synth_frames_collect_u8) */
/* the A <= 8 wide linear head of a Q network (nets.py:34-52's last nn.Linear) and the epsilon-greedy action in ONE launch:
 * q[n][a] = h[n] . w[a] + bias[a] from the last hidden activations h (N, H), w (A, H); action as trl_eps_greedy_i64 (u /
 * rand_act NULL: greedy); q_out (N, A) nullable.  H % 4 == 0, H <= 1024 (trl_dqn_act_supported), 16-byte aligned h, w */
int trl_dqn_act_supported(int H, int A);
int trl_dqn_act_f32(const float* h, const float* w, const float* bias, int N, int H, int A, const float* u,
                    const int64_t* rand_act, float epsilon, float* q_out, int64_t* action, int64_t* ring_row, int n_rows,
                    void* stream);     /* ring_row (nullable): ring_row[0] = (ring_row[0] + 1) % n_rows rides along, see
                                          trl_synth_frames_collect_u8 */
/* synthetic Atari-shaped env: (N, C, HW) uint8 frame stacks, Philox frames (see k_dqn.hip) */
int trl_synth_frames_step_u8(uint8_t* frames, const int64_t* acts, int32_t* t_env, int64_t env_seed_base,
                             int horizon, int A, uint8_t* next_obs, float* rewards, float* dones,
                             int N, int C, int HW, void* stream);
/* the same step filing the WHOLE transition into row ring_row[0] of the replay ring itself (bases of the (rows, N, ...)
 * tensors: uint8 obs / next_obs stacks, float acts / rewards / terminals / time_limits; base.py:22-28's keys) -- the
 * pre-step stacks are stored while they are shifted -- plus this step's rewards / dones (N) for the collector's bookkeeping.
 * The row lives on the device: a captured sequence of vector steps walks the ring, trl_synth_frames_reset_u8(ring_row)
 * advances it at the end of a step */
int trl_synth_frames_collect_u8(uint8_t* frames, const int64_t* acts, int32_t* t_env, int64_t env_seed_base, int horizon,
                                int A, uint8_t* ring_obs, uint8_t* ring_next_obs, float* ring_acts, float* ring_rewards,
                                float* ring_terminals, float* ring_time_limits, int64_t* ring_row, int n_rows,
                                float* step_rewards, float* step_dones, int32_t* cur_step, float* ep_return, int max_frames,
                                uint8_t* mask, double* epoch_reward, int32_t* ep_count, float* ep_log, int ep_cap, int step,
                                int N, int C, int HW, void* stream);
/* cur_step (nullable) .. step: trl_collector_bookkeep_f32's arguments -- the collector's bookkeeping of the step (collector/
 * base.py:199-228) and the reset of the envs that ended (their own stacks, vecenv.py:47-51) happen in the same launch;
 * the ring row is then advanced by the NEXT step's action launch (trl_dqn_act_f32) or by trl_synth_frames_reset_u8 */
/* ring_row (nullable): ring_row[0] = (ring_row[0] + 1) % n_rows rides along */
int trl_synth_frames_reset_u8(uint8_t* frames, int32_t* t_env, int64_t env_seed_base,
                              const uint8_t* mask, int64_t* ring_row, int n_rows, int N, int C, int HW, void* stream);

/* --- K6b: frame-deduplicating replay (torchrl/replay_buffers/memory_efficient_replay_buffer.py:5-33,
 * LazyFrames / FrameStack, torchrl/env/atari_wrapper.py:142-227).  stream: (S, N, HW) uint8 ring of single
 * frames per env; head[n]: monotone int32 position of env n's newest frame (slot = position % S).
 *   append  n_frames == 1: the newest frame (channel C-1) of stacks (N, C, HW); n_frames == C: the whole
 *           stack (episode start); mask (nullable) selects envs; head advances by n_frames
 *   gather  out (n_rows * N, C, HW): the k-stack whose newest frame sits at pos[row][n] + shift
 *           (shift 0 = obs, 1 = next_obs); *overrun |= 1 if a requested frame was already overwritten */
int trl_frame_stream_append_u8(const uint8_t* stacks, uint8_t* stream, int32_t* head, const uint8_t* mask,
                               int n_frames, int S, int N, int C, int HW, void* stream_);
int trl_frame_stream_gather_u8(const uint8_t* stream, const int32_t* pos, const int64_t* row_idx, int n_rows,
                               int shift, uint8_t* out, const int32_t* head, int32_t* overrun, int S, int N,
                               int C, int HW, void* stream_);

/* --- K1..K3 stand-alone: one VecOnPolicyCollector.take_actions as separate launches
 * (torchrl/collector/on_policy.py:90-155), for envs the persistent rollout kernel cannot carry
 * (a running observation normaliser needs all-env statistics before every policy forward).
 *   gauss_explore      act = [tanh](mean + exp(clamp(logstd)) * eps), logp = log pi(act) (eps / logp may be NULL)
 *   onpolicy_bookkeep  after env.step: epoch reward, running returns (logged + cleared on done),
 *                      rewards += discount * v_next * surpass (in place), terminals = reset_mask =
 *                      done | surpass, step counters, *any_flag |= any(reset_mask)   (:124-148)
 *   select_on_flag     out = *flag ? a : b -- partial_reset's whole-array return (:145-147) without a host sync
 *   select_on_mask     out = any(mask[0..N)) ? a : b -- the same for VecCollector.take_actions
 *                      (torchrl/collector/base.py:220-224), straight from the reset mask */
int trl_gauss_explore_f32(const float* mean, const float* logstd, const float* eps, float* act, float* logp,
                          int N, int A, int tanh_action, void* stream);
int trl_onpolicy_bookkeep_f32(float* rewards, const float* dones, const float* v_next, float discount,
                              float* terminals, int32_t* cur_step, float* ep_return, int max_episode_frames,
                              uint8_t* reset_mask, int32_t* any_flag, double* epoch_reward, int32_t* ep_count,
                              float* ep_log, int ep_cap, int step, int N, void* stream);
int trl_select_on_flag_f32(const int32_t* flag, const float* a, const float* b, float* out, int64_t n,
                           void* stream);
int trl_select_on_mask_f32(const uint8_t* mask, int N, const float* a, const float* b, float* out, int64_t n,
                           void* stream);

/* --- K18: running observation normaliser (torchrl/env/base_wrapper.py:44-121) --------------
 * state = {mean[D], var[D], count} fp64 on the device (Normalizer._mean/_var/_count; a fresh
 * normaliser is mean 0, var 1, count 1e-4, :64-69).  sums = {sum x [D], sum x^2 [D], n}.
 *   update_filt   one vector step in one launch: batch moments over the N rows, Chan merge
 *                 (update_mean_var_count, :44-60) when update != 0, then
 *                 out = clip((x - mean) / (sqrt(var) + 1e-4), +-clip) (filt, :86-89); out may be NULL
 *   batch_moments / merge / filt   the same three steps separately, so that env shards on several
 *                 GPUs can all-reduce (SUM) the batch moments between the first two.  D <= 64. */
int trl_norm_update_filt_f32(const float* x, double* state, float* out, int N, int D, float clip,
                             int update, void* stream);
int trl_norm_batch_moments_f64(const float* x, int N, int D, double* sums, void* stream);
int trl_norm_merge_f64(double* state, const double* sums, int D, void* stream);
int trl_norm_filt_f32(const float* x, const double* state, float* out, int N, int D, float clip,
                      void* stream);

/* --- host side of the reference's exploration-noise stream (torchrl/policies/distribution.py:60-76) ------------
 * The reference draws `Normal(0, 1).sample()` from the CPU torch generator = MT19937 (one engine call per float32 element
 * of a normal_() of n >= 16, n % 16 == 0 elements).  trl_mt19937_advance moves an engine state {state[624], left, next}
 * (the fields of torch's CPU generator state) forward by `calls` engine calls without producing outputs, so that the
 * state at any prefix of a block can be handed to another torch.Generator and several host threads draw the segments of
 * ONE torch.randn block concurrently, bit for bit.  Host code, no device work. */
int trl_mt19937_advance(uint32_t* state, int32_t* left, int64_t* next, int64_t calls);
/* K states of one stream in one pass: record k (state_bytes bytes at out + k * state_bytes) = the generator-state image
 * `tmpl` with its engine fields -- int32 `left` at off_left, int64 `next` at off_next, 624 x uint64 words at off_mt
 * (torch's CPU generator state) -- moved forward to engine call pos[k]; pos ascending, relative to tmpl.  What env shards
 * on several ranks need: rank r's rows of step t's (N_total, A) draw start at call t * N_total * A + r * N_local * A
 * (torchrl/policies/distribution.py:60-76 draws the tensor for ALL envs; torchrl/replay_buffers/on_policy.py:75-88 is what
 * makes the column blocks a partition of the reference's minibatch). */
int trl_mt19937_states_at(const uint8_t* tmpl, int64_t state_bytes, int64_t off_left, int64_t off_next, int64_t off_mt,
                          const int64_t* pos, int64_t K, uint8_t* out);
/* The same records (byte for byte) derived by `threads` host threads: the position list is cut into contiguous groups and
 * every group but the first starts from the template JUMPED ahead -- F^J = (x^J mod phi)(F) for MT19937's GF(2)-linear
 * transition F, ~0.2 ms per jump whatever J is, the polynomials cached per process (csrc/trl_mtjump.cpp).  At BASELINE
 * cfg 4 (8 ranks, 16 384 envs: 12.6 M engine calls per rollout between a rank's first and last chunk) the pass drops from
 * 2.4 ms of one thread to the time of one group.  trl_mt19937_jump_ready(): 1 once phi has been derived and checked
 * (Berlekamp-Massey on the recurrence, ~30 ms, first call); 0 = the _mt entry falls back to the sequential pass. */
int trl_mt19937_states_at_mt(const uint8_t* tmpl, int64_t state_bytes, int64_t off_left, int64_t off_next, int64_t off_mt,
                             const int64_t* pos, int64_t K, uint8_t* out, int threads);
int trl_mt19937_jump_ready(void);

/* --- calibration of the two rooflines (nominal AND achievable peaks) -------------------
 * No reference counterpart (the reference publishes no measurement); run by bench.py after its timed
 * region.  trl_peak_copy_f32: dst[0..n) = src[0..n) with 16-byte accesses on every CU (HBM bytes moved = 8 n); mode 0 / 1:
 * one 16 KB piece per workgroup with plain / non-temporal accesses, mode 2: persistent grid-stride (the caller quotes the best),
 * mode 3: 4-byte accesses (calibrates the profiler's FETCH_SIZE / WRITE_SIZE for the gradient kernel's access width).
 * trl_peak_mfma_f32: `workgroups` x 4 waves each issue `iters` x 4 independent v_mfma_f32_32x32x2_f32 from registers
 * (FLOPs = workgroups * 4 * iters * 4 * 4096); out: workgroups * 256 floats (sink). */
int trl_peak_copy_f32(const float* src, float* dst, int64_t n, int mode, void* stream);
int trl_peak_mfma_f32(float* out, int workgroups, int iters, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TRL_HIP_H */
