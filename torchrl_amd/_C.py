"""ctypes binding of libtrl_hip.so (include/trl_hip.h).

There is no CPU or pure-PyTorch fallback: every op below either runs the HIP
kernel on the tensors' device or raises.  The library is loaded lazily so that
host-only logic (index streams, ring bookkeeping, argument checks) imports on a
machine without the .so; the first kernel call then fails loudly.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TRL_LIB") or os.path.join(_HERE, "lib", "libtrl_hip.so")   # TRL_LIB: experimental builds
_lib = None

ACT_TANH, ACT_RELU, ACT_NONE = 0, 1, 2
ACT_CODES = {"tanh": ACT_TANH, "relu": ACT_RELU}

c_float_p = C.POINTER(C.c_float)
c_double_p = C.POINTER(C.c_double)
c_i64_p = C.POINTER(C.c_int64)
c_i32_p = C.POINTER(C.c_int32)
c_u8_p = C.POINTER(C.c_uint8)


# ---- leaving the HIP path must be loud ----
# Product classes keep the reference's nn.Module / torch.distributions protocol for callers outside the hot path
# (autograd through a network, CPU tensors in unit tests of host logic).  Whenever a CUDA tensor takes such a route the
# site is counted here and logged once at WARNING; TRL_STRICT=1 turns it into an error.  tests/test_no_eager_gpu.py
# asserts that whole PPO / SAC / DQN epochs leave the counter at zero.
EAGER_FALLBACKS = {}


def note_eager(site, why=""):
    EAGER_FALLBACKS[site] = EAGER_FALLBACKS.get(site, 0) + 1
    if os.environ.get("TRL_STRICT") == "1":
        raise TrlError("%s left the HIP path (%s) and TRL_STRICT=1" % (site, why))
    if EAGER_FALLBACKS[site] == 1:
        import logging
        logging.getLogger("torchrl_amd").warning("%s: a CUDA tensor takes the eager PyTorch route (%s); "
                                                 "this is outside the HIP hot path", site, why)


def eager_fallback_count():
    return sum(EAGER_FALLBACKS.values())


class TrlError(RuntimeError):
    pass


class RolloutArgs(C.Structure):
    _fields_ = [
        ("pf_params", C.c_void_p), ("vf_params", C.c_void_p),
        ("D", C.c_int), ("H", C.c_int), ("A", C.c_int), ("act", C.c_int),
        ("tanh_action", C.c_int),
        ("env_A", C.c_void_p), ("env_B", C.c_void_p),
        ("reward_scale", C.c_float), ("horizon", C.c_int), ("env_seed_base", C.c_int64),
        ("cur_obs", C.c_void_p), ("t_env", C.c_void_p), ("cur_step", C.c_void_p),
        ("episode_idx", C.c_void_p), ("ep_return", C.c_void_p),
        ("noise", C.c_void_p), ("noise_step0", C.c_int64), ("deterministic", C.c_int),
        ("obs", C.c_void_p), ("next_obs", C.c_void_p), ("acts", C.c_void_p), ("values", C.c_void_p),
        ("rewards", C.c_void_p), ("terminals", C.c_void_p), ("time_limits", C.c_void_p),
        ("old_logp", C.c_void_p),
        ("rows", C.c_int), ("top", C.c_int), ("N", C.c_int), ("n_steps", C.c_int),
        ("max_episode_frames", C.c_int), ("discount", C.c_float),
        ("epoch_reward", C.c_void_p), ("ep_count", C.c_void_p), ("ep_log", C.c_void_p),
        ("ep_cap", C.c_int), ("step0", C.c_int),
        ("norm_state", C.c_void_p), ("policy_obs", C.c_void_p), ("norm_workspace", C.c_void_p),
        ("norm_clip", C.c_float), ("norm_update", C.c_int), ("normalize_partial_reset", C.c_int),
        ("clear_header", C.c_void_p), ("boot_values", C.c_void_p),
        ("publish_dst", C.c_void_p), ("publish_src", C.c_void_p), ("publish_words", C.c_int64),
        ("noise_flag", C.c_void_p), ("noise_stamp", C.c_uint32),
        ("stage_src", C.c_void_p), ("stage_dst", C.c_void_p), ("stage_n", C.c_int64),
        ("stage_ready", C.c_void_p), ("stage_job", C.c_uint32), ("stage_state", C.c_void_p), ("stage_ack", C.c_void_p),
        ("value_wait_event", C.c_void_p),
    ]


LOSS_PPO_CLIP, LOSS_A2C = 0, 1                                   # TRL_LOSS_* of include/trl_hip.h


class PpoBatchArgs(C.Structure):
    _fields_ = [
        ("obs", C.c_void_p), ("acts", C.c_void_p), ("advs", C.c_void_p), ("rets", C.c_void_p),
        ("old_values", C.c_void_p), ("old_logp", C.c_void_p),
        ("row_idx", C.c_void_p), ("rows_mb", C.c_int), ("N", C.c_int),
        ("adv_raw", C.c_void_p), ("n_global", C.c_double),
        ("pf_params", C.c_void_p), ("vf_params", C.c_void_p),
        ("D", C.c_int), ("H", C.c_int), ("A", C.c_int), ("act", C.c_int),
        ("clip_para", C.c_float), ("entropy_coeff", C.c_float),
        ("clipped_value_loss", C.c_int), ("tanh_action", C.c_int), ("loss_mode", C.c_int),
        ("partial", C.c_void_p), ("scal_partial", C.c_void_p), ("n_wg", C.c_int), ("n_wg_pf", C.c_int),
    ]


class AdamArgs(C.Structure):
    _fields_ = [
        ("params", C.c_void_p), ("grads", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
        ("n_groups", C.c_int), ("group_sizes", C.c_int * 4), ("group_lr", C.c_float * 4),
        ("max_norm", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
        ("step_count", C.c_int), ("grad_scale", C.c_float), ("norms_out", C.c_void_p),
        ("device_state", C.c_int), ("step_state", C.c_void_p), ("device_lr", C.c_void_p),
    ]


class ConvRiders(C.Structure):          # include/trl_hip.h trl_conv_riders_t
    _fields_ = [("n_perm", C.c_int), ("perm_src", C.c_void_p * 4), ("perm_dst", C.c_void_p * 4), ("perm_cout", C.c_int * 4),
                ("perm_c", C.c_int * 4), ("perm_khw", C.c_int * 4), ("n_dx", C.c_int), ("dx_w", C.c_void_p * 4),
                ("dx_ws", C.c_void_p * 4), ("dx_cin", C.c_int * 4), ("dx_cout", C.c_int * 4), ("dx_kh", C.c_int * 4),
                ("dx_kw", C.c_int * 4), ("dx_sh", C.c_int * 4), ("dx_sw", C.c_int * 4)]


# name -> (restype, argtypes); the loader checks every one of these symbols exists
SIGNATURES = {
    "trl_last_error": (C.c_char_p, []),
    "trl_abi_version": (C.c_int, []),
    "trl_gae_f32": (C.c_int, [C.c_void_p] * 8 + [C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p]),
    "trl_discount_reward_f32": (C.c_int, [C.c_void_p] * 8 + [C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "trl_gather_rows_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p]),
    "trl_gather_rows_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p]),
    "trl_gather_rows_multi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_int64, C.c_void_p]),
    "trl_adv_stats_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "trl_ppo_epoch_prologue_workspace": (C.c_int64, [C.c_int]),
    "trl_ppo_epoch_prologue_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "trl_mlp2_forward_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "trl_rollout_synth_f32": (C.c_int, [C.POINTER(RolloutArgs), C.c_void_p]),
    "trl_rollout_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "trl_stage_h2d_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_uint32, C.c_void_p]),
    "trl_ppo_partial_stride": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "trl_mlp2_forward_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "trl_ppo_minibatch_grad_f32": (C.c_int, [C.POINTER(PpoBatchArgs), C.c_void_p]),
    "trl_ppo_wg_split": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "trl_rollout_norm_workspace": (C.c_int, [C.c_int]),
    "trl_rollout_norm_max_envs": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "trl_frame_stream_append_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "trl_frame_stream_gather_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "trl_detac_losses_f32": (C.c_int, [C.c_void_p] * 7 + [C.c_float, C.c_int] + [C.c_void_p] * 5),
    "trl_noisy_action_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]),
    "trl_gauss_explore_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "trl_onpolicy_bookkeep_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "trl_select_on_flag_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "trl_comm_unique_id_bytes": (C.c_int, []),
    "trl_comm_peer_handle_bytes": (C.c_int, []),
    "trl_comm_max_ranks": (C.c_int, []),
    "trl_comm_get_unique_id": (C.c_int, [C.c_void_p]),
    "trl_comm_init": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p]),
    "trl_comm_peer_export": (C.c_int, [C.c_void_p, C.c_void_p]),
    "trl_comm_peer_open": (C.c_int, [C.c_void_p, C.c_void_p]),
    "trl_comm_peer_ready": (C.c_int, [C.c_void_p]),
    "trl_comm_peer_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "trl_comm_has_rccl": (C.c_int, [C.c_void_p]),
    "trl_comm_error": (C.c_int, [C.c_void_p]),
    "trl_comm_error_peek": (C.c_int, [C.c_void_p]),
    "trl_comm_error_detail": (C.c_int, [C.c_void_p, C.c_void_p]),
    "trl_comm_link_info": (C.c_int, [C.c_int, C.c_int, C.c_void_p]),
    "trl_comm_set_wait_footprint": (C.c_int, [C.c_void_p, C.c_int]),
    "trl_comm_peer_buffer_kind": (C.c_int, [C.c_void_p]),
    "trl_comm_destroy": (C.c_int, [C.c_void_p]),
    "trl_allreduce_sum_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "trl_allreduce_f64": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p]),
    "trl_ppo_reduce_adam_xrank_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                C.c_void_p, C.POINTER(AdamArgs), C.c_void_p, C.c_void_p, C.c_void_p]),
    "trl_select_on_mask_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "trl_norm_update_filt_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "trl_norm_batch_moments_f64": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "trl_norm_merge_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "trl_norm_filt_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "trl_ppo_reduce_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "trl_clip_adam_f32": (C.c_int, [C.POINTER(AdamArgs), C.c_void_p]),
    "trl_clip_adam_polyak_f32": (C.c_int, [C.POINTER(AdamArgs), C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_int,
                                          C.c_void_p, C.c_int, C.c_void_p]),
    "trl_ppo_reduce_adam_workspace": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "trl_ppo_reduce_adam_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                          C.POINTER(AdamArgs), C.c_void_p, C.c_void_p]),
    "trl_ppo_reduce_adam_net_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                              C.POINTER(AdamArgs), C.c_void_p, C.c_void_p]),
    "trl_ppo_reduce_adam_xrank_net_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                    C.c_void_p, C.POINTER(AdamArgs), C.c_void_p, C.c_void_p, C.c_void_p]),
    "trl_ppo_step_workspace": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "trl_ppo_step_max_workgroups": (C.c_int, []),
    "trl_ppo_minibatch_step_f32": (C.c_int, [C.POINTER(PpoBatchArgs), C.c_void_p, C.c_void_p, C.POINTER(AdamArgs), C.c_void_p, C.c_void_p]),
    "trl_synth_reset_f32": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_int64, C.c_void_p]),
    "trl_gauss_logp_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "trl_concat2_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p]),
    "trl_tanh_gauss_rsample_fwd_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p]),
    "trl_tanh_gauss_rsample_bwd_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_float] * 3 + [C.c_void_p] + [C.c_int] * 3 + [C.c_void_p]),
    "trl_tanh_gauss_rsample_bwd_cols_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 2 + [C.c_void_p] + [C.c_float] * 3 +
                                            [C.c_void_p] + [C.c_int] * 3 + [C.c_void_p]),
    "trl_fold_clip_adam_polyak_workspace": (C.c_int, []),
    "trl_fold_clip_adam_polyak_f32": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(AdamArgs), C.c_void_p,
                                              C.c_int64, C.c_int64, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                              C.c_void_p, C.c_void_p]),
    "trl_fold_scope_begin": (C.c_int, []),
    "trl_fold_scope_end": (C.c_int, [C.c_void_p]),
    "trl_sac_policy_grad_supported": (C.c_int, [C.c_int, C.c_int]),
    "trl_sac_policy_grad_f32": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int] +
                                [C.c_void_p] * 4 + [C.c_float] * 3 + [C.c_void_p, C.c_int, C.c_int, C.c_int] +
                                [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "trl_sac_samples_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int64] + [C.c_void_p] * 10 + [C.c_int] * 4 +
                            [C.c_void_p, C.c_void_p]),
    "trl_moments_multi_f64": (C.c_int, [C.c_int] + [C.c_void_p] * 9 + [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "trl_sac_losses_f32": (C.c_int, [C.c_void_p] * 11 + [C.c_float, C.c_int] + [C.c_void_p] * 7 + [C.c_float] * 5 +
                           [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "trl_synth_collect_step_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int64, C.c_int64, C.c_int] + [C.c_void_p] * 6 +
                                   [C.c_float, C.c_int, C.c_int, C.c_int64] + [C.c_void_p] * 6 + [C.c_int, C.c_void_p] +
                                   [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p]),
    "trl_collector_bookkeep_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p]),
    "trl_sac_alpha_step_f32": (C.c_int, [C.c_void_p, C.c_int] + [C.c_float] * 5 + [C.c_void_p] * 3),
    "trl_slice_add_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p]),
    "trl_polyak_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p]),
    "trl_philox_normal_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    "trl_synth_env_step_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_float, C.c_int] + [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p]),
    "trl_im2col_f32": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 8 + [C.c_void_p]),
    "trl_im2col_u8_nchw": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 8 + [C.c_float, C.c_float, C.c_void_p]),
    "trl_col2im_f32": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 8 + [C.c_void_p]),
    "trl_conv_bwd_input_nhwc_ok": (C.c_int, [C.c_int] * 6),
    "trl_conv_bwd_input_nhwc_workspace": (C.c_int, [C.c_int] * 4),
    "trl_conv_bwd_input_nhwc_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                             C.c_void_p] + [C.c_int] * 10 + [C.c_void_p]),
    "trl_conv_bwd_input_nhwc_prep_f32": (C.c_int, [C.c_int] + [C.c_void_p] * 8 + [C.c_void_p]),
    "trl_transpose_bpc_gate_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p] + [C.c_int] * 3
                                   + [C.c_void_p]),
    "trl_transpose_bpc_f32": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 3 + [C.c_void_p]),
    "trl_conv_fwd_u8_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 8 + [C.c_float, C.c_float, C.c_int, C.c_int] +
                            [C.POINTER(ConvRiders), C.c_void_p]),
    "trl_conv_fwd_u8_pair_f32": (C.c_int, [C.c_void_p] * 8 + [C.c_int] * 8 + [C.c_float, C.c_float, C.c_int, C.c_int] +
                                 [C.POINTER(ConvRiders), C.c_void_p]),
    "trl_conv_bwd_weight_workspace": (C.c_int, [C.c_int] * 9),
    "trl_conv_fwd_nhwc_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 12 + [C.c_void_p]),
    "trl_conv_bwd_weight_nhwc_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 9
                                     + [C.c_void_p]),
    "trl_conv_bwd_weight_u8_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 8
                                   + [C.c_float, C.c_float, C.c_int, C.c_void_p]),
    "trl_dqn_td_loss_f32": (C.c_int, [C.c_void_p] * 6 + [C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_void_p, C.c_void_p]),
    "trl_dqn_head_supported": (C.c_int, [C.c_int, C.c_int]),
    "trl_dqn_head_workspace": (C.c_int64, [C.c_int, C.c_int]),
    "trl_dqn_head_f32": (C.c_int, [C.c_void_p] * 10 + [C.c_float, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 7 +
                         [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "trl_quantile_huber_f32": (C.c_int, [C.c_void_p] * 6 + [C.c_float, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 4 +
                               [C.c_int, C.c_void_p, C.c_void_p]),
    "trl_dqn_act_supported": (C.c_int, [C.c_int, C.c_int]),
    "trl_dqn_act_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "trl_eps_greedy_i64": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_void_p]),
    "trl_synth_frames_step_u8": (C.c_int, [C.c_void_p] * 3 + [C.c_int64, C.c_int, C.c_int] + [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p]),
    "trl_synth_frames_reset_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]),
    "trl_synth_frames_collect_u8": (C.c_int, [C.c_void_p] * 3 + [C.c_int64, C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_int] +
                                    [C.c_void_p] * 2 + [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                        C.c_void_p, C.c_int, C.c_int] + [C.c_int] * 3 + [C.c_void_p]),
    "trl_linear_fwd_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "trl_mt19937_advance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "trl_mt19937_states_at": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "trl_mt19937_states_at_mt": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                          C.c_int]),
    "trl_mt19937_jump_ready": (C.c_int, []),
    "trl_peak_copy_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "trl_peak_mfma_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "trl_adv_normalize_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "trl_trpo_surrogate_workspace": (C.c_int, [C.c_int] * 2),
    "trl_trpo_surrogate_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_float] + [C.c_void_p] * 5),
    "trl_jvp_gate_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "trl_fisher_scale_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "trl_ratio_loss_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int, C.c_void_p, C.c_void_p]),
    "trl_mse_value_loss_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "trl_vmpo_losses_workspace": (C.c_int, [C.c_int] * 2),
    "trl_vmpo_losses_f32": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 3 + [C.c_float] * 3 + [C.c_void_p] * 5),
    "trl_ppo_generic_losses_workspace": (C.c_int, [C.c_int] * 2),
    "trl_ppo_generic_losses_f32": (C.c_int, [C.c_void_p] * 9 + [C.c_double, C.c_int, C.c_int, C.c_float, C.c_float,
                                                                 C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6),
    "trl_linear_fwd_workspace": (C.c_int, [C.c_int] * 3),
    "trl_linear_fwd_group_f32": (C.c_int, [C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]),
    "trl_linear_bwd_input_group_f32": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
                                       + [C.c_int] * 3 + [C.c_void_p]),
    "trl_linear_bwd_weight_group_f32": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 4
                                        + [C.c_int] * 3 + [C.c_void_p]),
    "trl_outer_gate_group_f32": (C.c_int, [C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p]),
    "trl_mlp3_forward_ok": (C.c_int, [C.c_int] * 4),
    "trl_mlp3_forward_group_f32": (C.c_int, [C.c_int] + [C.c_void_p] * 10 + [C.c_int] * 5 + [C.c_void_p]),
    "trl_linear_fwd_splitk_group_f32": (C.c_int, [C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p, C.c_void_p]),
    "trl_conv_fwd_nhwc_group_f32": (C.c_int, [C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 12 + [C.c_void_p]),
    "trl_linear_fwd_splitk_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p, C.c_void_p]),
    "trl_linear_bwd_input_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "trl_linear_bwd_input_workspace": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "trl_linear_bwd_input_splitk_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "trl_linear_bwd_weight_workspace": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "trl_linear_bwd_weight_splits": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "trl_linear_bwd_weight_partials_group_f32": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                                          C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "trl_fold_partials_multi_f32": (C.c_int, [C.c_int] + [C.c_void_p] * 4 + [C.c_void_p]),
    "trl_linear_bwd_weight_multi_splits": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "trl_linear_bwd_weight_partials_multi_f32": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                          C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "trl_linear_bwd_weight_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
}


def lib():
    """Load (once) and type the shared library; raise if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TrlError(
                "libtrl_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `python torchrl_amd/build.py`. torchrl_amd has no CPU fallback." % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        lax = os.environ.get("TRL_LIB_LAX") == "1"   # tools/ A/B runs against an older experimental build (TRL_LIB)
        for name, (res, args) in SIGNATURES.items():
            if lax and not hasattr(handle, name):
                continue
            fn = getattr(handle, name)          # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().trl_last_error().decode() or "error %d" % code
        raise TrlError("%s failed (%d): %s" % (what, code, msg))


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def dev_ptr(t, dtype=torch.float32, name="tensor", allow_none=False):
    """Pointer to a contiguous device tensor of the given dtype (loud on anything else)."""
    if t is None:
        if allow_none:
            return C.c_void_p(0)
        raise TrlError("%s is None" % name)
    if not isinstance(t, torch.Tensor):
        raise TrlError("%s must be a torch.Tensor, got %s" % (name, type(t).__name__))
    if t.device.type != "cuda":
        raise TrlError("%s lives on %s: torchrl_amd kernels need a GPU tensor (there is no CPU path)"
                       % (name, t.device))
    if t.dtype != dtype:
        raise TrlError("%s has dtype %s, expected %s" % (name, t.dtype, dtype))
    if not t.is_contiguous():
        raise TrlError("%s must be contiguous" % name)
    return C.c_void_p(t.data_ptr())


def capture_graph(launches):
    """`launches()` captured into a HIP graph; returns (graph, what launches() returned).  The Python garbage collector is
    paused meanwhile: a collection that frees a page-locked tensor makes torch's host allocator query its events, which
    HIP refuses while a stream is capturing (seen as an abort of the process in the middle of a capture)."""
    import gc
    graph = torch.cuda.CUDAGraph()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph):
            out = launches()
    finally:
        if was_enabled:
            gc.enable()
    return graph, out


class PinnedStager:
    """Small host -> device uploads by a host that runs AHEAD of the device (replayed graphs: whole epochs).  An
    asynchronous copy reads its page-locked source when the stream gets there, not when it is issued, so a single staging
    buffer rewritten for the next upload races with the previous one.  `depth` buffers in rotation, each with the event of
    its last copy: `stage()` hands out the next buffer once that copy has been executed (normally long ago), `upload(dst)`
    issues the copy and records the event.  The buffers live as long as the stager (nothing page-locked is ever freed
    while a stream captures, see capture_graph)."""

    def __init__(self, nelem, dtype, depth=4):
        pin = torch.cuda.is_available()
        self._bufs = [torch.zeros(int(nelem), dtype=dtype) for _ in range(depth)]
        if pin:
            self._bufs = [b.pin_memory() for b in self._bufs]
        self._events, self._k = [None] * depth, 0

    def stage(self):
        ev = self._events[self._k]
        if ev is not None:
            ev.synchronize()
        return self._bufs[self._k]

    def upload(self, dst):
        k = self._k
        dst.copy_(self._bufs[k], non_blocking=True)
        if dst.is_cuda:
            ev = self._events[k] or torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dst.device))
            self._events[k] = ev
        self._k = (k + 1) % len(self._bufs)
        return dst


# ------------------------------------------------------------------ thin op wrappers
def gae(rewards, values, terminals, time_limits, last_value, advs, rets, gamma, tau, tl_filter,
        last_terminal=None):
    T, N = rewards.shape[0], rewards.shape[1]
    check(lib().trl_gae_f32(dev_ptr(rewards, name="rewards"), dev_ptr(values, name="values"),
                            dev_ptr(terminals, name="terminals"),
                            dev_ptr(time_limits, name="time_limits", allow_none=not tl_filter),
                            dev_ptr(last_value, name="last_value"),
                            dev_ptr(last_terminal, name="last_terminal", allow_none=True),
                            dev_ptr(advs, name="advs"),
                            dev_ptr(rets, name="rets"), T, N, float(gamma), float(tau), int(bool(tl_filter)),
                            stream_ptr(rewards.device)), "trl_gae_f32")


def discount_reward(rewards, values, terminals, time_limits, last_value, advs, rets, gamma, tl_filter,
                    last_terminal=None):
    T, N = rewards.shape[0], rewards.shape[1]
    check(lib().trl_discount_reward_f32(dev_ptr(rewards, name="rewards"), dev_ptr(values, name="values"),
                                        dev_ptr(terminals, name="terminals"),
                                        dev_ptr(time_limits, name="time_limits", allow_none=not tl_filter),
                                        dev_ptr(last_value, name="last_value"),
                                        dev_ptr(last_terminal, name="last_terminal", allow_none=True),
                                        dev_ptr(advs, name="advs"),
                                        dev_ptr(rets, name="rets"), T, N, float(gamma), int(bool(tl_filter)),
                                        stream_ptr(rewards.device)), "trl_discount_reward_f32")


def gather_rows(src, row_idx, out=None):
    """out[i] = src[row_idx[i]] over the leading (time-row) dimension."""
    n = int(row_idx.numel())
    if out is None:
        out = torch.empty((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    row_elems = 1
    for s in src.shape[1:]:
        row_elems *= int(s)
    if src.dtype == torch.float32:
        fn, name = lib().trl_gather_rows_f32, "trl_gather_rows_f32"
    elif src.dtype == torch.uint8:
        fn, name = lib().trl_gather_rows_u8, "trl_gather_rows_u8"
    else:
        raise TrlError("gather_rows: dtype %s not supported" % src.dtype)
    check(fn(dev_ptr(src, src.dtype, "src"), dev_ptr(row_idx, torch.int64, "row_idx"),
             dev_ptr(out, src.dtype, "out"), n, row_elems, int(src.shape[0]), stream_ptr(src.device)), name)
    return out


def gather_rows_multi(srcs, row_idx, outs, slab_counter=None, n_rows=None):
    """outs[k][i] = srcs[k][row_idx[i]] for every key k in one launch (all srcs share the leading row count).
    With `slab_counter` (a device float64 update counter) `row_idx` is an index SLAB {first, sets, idx[sets][n_rows]} and
    the set (counter - first) is gathered."""
    n, k = int(row_idx.numel()) if slab_counter is None else int(n_rows), len(srcs)
    rows = int(srcs[0].shape[0])
    if any(int(s.shape[0]) != rows for s in srcs):
        raise TrlError("gather_rows_multi: keys with different row counts")
    sp, dp, nb = (C.c_void_p * k)(), (C.c_void_p * k)(), (C.c_int64 * k)()
    for j, (s, o) in enumerate(zip(srcs, outs)):
        if s.dtype != o.dtype or o.numel() != n * s[0].numel():
            raise TrlError("gather_rows_multi: out[%d] does not match the batch" % j)
        sp[j], dp[j] = dev_ptr(s, s.dtype, "src"), dev_ptr(o, o.dtype, "out")
        nb[j] = s[0].numel() * s.element_size()
    check(lib().trl_gather_rows_multi(sp, dp, nb, k, dev_ptr(row_idx, torch.int64, "row_idx"),
                                      dev_ptr(slab_counter, torch.float64, "counter", allow_none=True), n, rows,
                                      stream_ptr(srcs[0].device)), "trl_gather_rows_multi")
    return outs


def adv_stats(advs, row_idx_2d, raw_out):
    n_mb, rows_mb = int(row_idx_2d.shape[0]), int(row_idx_2d.shape[1])
    check(lib().trl_adv_stats_f64(dev_ptr(advs, name="advs"), dev_ptr(row_idx_2d, torch.int64, "row_idx"),
                                  n_mb, rows_mb, int(advs.shape[1]),
                                  dev_ptr(raw_out, torch.float64, "raw_out"), stream_ptr(advs.device)),
          "trl_adv_stats_f64")
    return raw_out


def stage_h2d(host, dev, state, stamp, stream):
    """`host` (page-locked float32) -> `dev` by a kernel on `stream` (a torch.cuda.Stream); state[0] = stamp (int32 device
    tensor of 2, zeroed once) tells consumers on other streams that the block has landed."""
    if not (host.is_pinned() and host.dtype == torch.float32 and host.is_contiguous() and host.numel() == dev.numel()):
        raise TrlError("stage_h2d: a contiguous page-locked float32 source of the destination's size is required")
    check(lib().trl_stage_h2d_f32(C.c_void_p(host.data_ptr()), dev_ptr(dev, name="dev"), host.numel(),
                                  dev_ptr(state, torch.int32, "state"), int(stamp) & 0xFFFFFFFF,
                                  C.c_void_p(stream.cuda_stream)), "trl_stage_h2d_f32")


def ppo_epoch_prologue_workspace(n_mb, device):
    """Zeroed workspace of `ppo_epoch_prologue` (keep it: it carries arrival counters from launch to launch)."""
    return torch.zeros(int(lib().trl_ppo_epoch_prologue_workspace(int(n_mb))), dtype=torch.uint8, device=device)


def _mapped_ptr(t, name):
    """Device-readable address of a tensor: device memory, or page-locked host memory (which the GPU reads in place)."""
    if not isinstance(t, torch.Tensor) or not t.is_contiguous():
        raise TrlError("%s must be a contiguous tensor" % name)
    if t.device.type != "cuda" and not t.is_pinned():
        raise TrlError("%s lives in pageable host memory: the device cannot read it" % name)
    return C.c_void_p(t.data_ptr())


def ppo_epoch_prologue(advs, row_idx_2d, raw_out, workspace, zero=None, copies=()):
    """`adv_stats` over sliced minibatches (>= 256 workgroups) with the clearing of `zero` (float64 tensor) and up to 4
    copies `(dst, src)` (same byte size, a multiple of 4) riding in the same launch (include/trl_hip.h K7).  row_idx_2d
    and the copy sources may be page-locked host tensors."""
    n_mb, rows_mb = int(row_idx_2d.shape[0]), int(row_idx_2d.shape[1])
    if row_idx_2d.dtype != torch.int64:
        raise TrlError("row_idx must be int64")
    n = len(copies)
    dsts, srcs, words = (C.c_void_p * max(n, 1))(), (C.c_void_p * max(n, 1))(), (C.c_int64 * max(n, 1))()
    for q, (dst, src) in enumerate(copies):
        nb = dst.numel() * dst.element_size()
        if nb != src.numel() * src.element_size() or nb % 4:
            raise TrlError("ppo_epoch_prologue: copy %d needs two tensors of one byte size, a multiple of 4" % q)
        dsts[q], srcs[q], words[q] = dev_ptr(dst, dst.dtype, "copy_dst").value, _mapped_ptr(src, "copy_src").value, nb // 4
    check(lib().trl_ppo_epoch_prologue_f64(
        dev_ptr(advs, name="advs"), _mapped_ptr(row_idx_2d, "row_idx"), n_mb, rows_mb, int(advs.shape[1]),
        dev_ptr(raw_out, torch.float64, "raw_out"), dev_ptr(workspace, torch.uint8, "workspace", allow_none=True),
        dev_ptr(zero, torch.float64, "zero", allow_none=True), 0 if zero is None else zero.numel(),
        n, dsts, srcs, words, stream_ptr(advs.device)), "trl_ppo_epoch_prologue_f64")
    return raw_out


def mlp2_forward(params, x, D, H, O, act, out=None):
    M = int(x.shape[0])
    if out is None:
        out = torch.empty((M, O), dtype=torch.float32, device=x.device)
    check(lib().trl_mlp2_forward_f32(dev_ptr(params, name="params"), dev_ptr(x, name="x"),
                                     dev_ptr(out, name="out"), M, D, H, O, act, stream_ptr(x.device)),
          "trl_mlp2_forward_f32")
    return out


def rollout(args, device):
    check(lib().trl_rollout_synth_f32(C.byref(args), stream_ptr(device)), "trl_rollout_synth_f32")


def ppo_partial_stride(D, H, A):
    ps = lib().trl_ppo_partial_stride(D, H, A)
    if ps < 0:
        check(ps, "trl_ppo_partial_stride")
    return ps


def ppo_minibatch_grad(args, device):
    check(lib().trl_ppo_minibatch_grad_f32(C.byref(args), stream_ptr(device)), "trl_ppo_minibatch_grad_f32")


def ppo_reduce(partial, scal_partial, n_wg, D, H, A, grads, info, pf_params=None, n_wg_pf=0):
    check(lib().trl_ppo_reduce_f32(dev_ptr(partial, name="partial"),
                                   dev_ptr(scal_partial, torch.float64, "scal_partial"), n_wg, n_wg_pf, D, H, A,
                                   dev_ptr(pf_params, name="pf_params", allow_none=True),
                                   dev_ptr(grads, name="grads"), dev_ptr(info, torch.float64, "info"),
                                   stream_ptr(partial.device)), "trl_ppo_reduce_f32")


def clip_adam(args, device):
    check(lib().trl_clip_adam_f32(C.byref(args), stream_ptr(device)), "trl_clip_adam_f32")


def clip_adam_polyak(args, target, source, tau, device, file=None):
    """clip + Adam, then target <- (1 - tau) target + tau source (the Polyak kernel also advances the device step state).
    file = (raw uint8 statistics block, ring (slots, raw bytes) uint8): the Polyak launch also archives `raw` into ring row
    (steps taken before this update) % slots."""
    raw, ring = file if file is not None else (None, None)
    if ring is not None and (ring.dim() != 2 or int(ring.shape[1]) != raw.numel() or not ring.is_contiguous()):
        raise TrlError("clip_adam_polyak: ring rows must be statistics blocks")
    check(lib().trl_clip_adam_polyak_f32(C.byref(args), dev_ptr(target, name="target"), dev_ptr(source, name="source"),
                                         int(target.numel()), float(tau), dev_ptr(raw, torch.uint8, "raw", allow_none=True),
                                         0 if raw is None else int(raw.numel()),
                                         dev_ptr(ring, torch.uint8, "ring", allow_none=True),
                                         0 if ring is None else int(ring.shape[0]), stream_ptr(device)),
          "trl_clip_adam_polyak_f32")


def fold_clip_adam_polyak_workspace(device):
    return torch.zeros(int(lib().trl_fold_clip_adam_polyak_workspace()), dtype=torch.uint8, device=device)


def synth_collect_step(env, head, eps, cur_step, ep_return, max_frames, rows, mask, epoch_reward, ep_count, ep_log, step,
                       tanh_action, noise=None):
    """One off-policy vector step on the synthetic env in one launch; rows = (obs, acts, next_obs, rewards, terminals,
    time_limits) destination rows (obs / acts / time_limits may be None).  eps None: `noise` = (seed, counter, first row)
    of the device Philox draw, generated inside the launch."""
    N, D, A = int(env.cur_obs.shape[0]), int(env.cur_obs.shape[1]), int(head.shape[1]) // 2
    obs_row, acts_row, next_row, rew_row, done_row, tl_row = rows
    seed, ctr, row0 = noise if eps is None else (0, 0, 0)
    check(lib().trl_synth_collect_step_f32(
        dev_ptr(env.cur_obs, name="cur_obs"), dev_ptr(head, name="head"), dev_ptr(eps, name="eps", allow_none=True),
        int(seed), int(ctr), int(row0), dev_ptr(env.env_A, name="env_A"), dev_ptr(env.env_B, name="env_B"), dev_ptr(env.t_env, torch.int32, "t_env"),
        dev_ptr(cur_step, torch.int32, "cur_step"), dev_ptr(env.episode_idx, torch.int32, "episode_idx"),
        dev_ptr(ep_return, name="ep_return"), float(env.effective_reward_scale), int(env.horizon), int(max_frames),
        int(env.seed_base), dev_ptr(obs_row, name="obs_row", allow_none=True),
        dev_ptr(acts_row, name="acts_row", allow_none=True), dev_ptr(next_row, name="next_row"),
        dev_ptr(rew_row, name="rew_row"), dev_ptr(done_row, name="done_row"),
        dev_ptr(tl_row, name="tl_row", allow_none=True), 0, None, dev_ptr(mask, torch.uint8, "mask"),
        dev_ptr(epoch_reward, torch.float64, "epoch_reward"), dev_ptr(ep_count, torch.int32, "ep_count"),
        dev_ptr(ep_log, name="ep_log"), int(ep_log.shape[0]), int(step), N, D, A, int(bool(tanh_action)),
        stream_ptr(head.device)), "trl_synth_collect_step_f32")


def synth_collect_step_dyn(env, head, cur_step, ep_return, max_frames, ring, state, mask, epoch_reward, ep_count, ep_log,
                           tanh_action, noise_seed, noise_row0):
    """`synth_collect_step` with the step counter / ring row / epoch start on the device (`state`, 4 int64): graph-replayable.
    ring = the six whole ring tensors (obs, acts, next_obs, rewards, terminals, time_limits)."""
    N, D, A = int(env.cur_obs.shape[0]), int(env.cur_obs.shape[1]), int(head.shape[1]) // 2
    check(lib().trl_synth_collect_step_f32(
        dev_ptr(env.cur_obs, name="cur_obs"), dev_ptr(head, name="head"), None, int(noise_seed), 0, int(noise_row0),
        dev_ptr(env.env_A, name="env_A"), dev_ptr(env.env_B, name="env_B"), dev_ptr(env.t_env, torch.int32, "t_env"),
        dev_ptr(cur_step, torch.int32, "cur_step"), dev_ptr(env.episode_idx, torch.int32, "episode_idx"),
        dev_ptr(ep_return, name="ep_return"), float(env.effective_reward_scale), int(env.horizon), int(max_frames),
        int(env.seed_base), *[dev_ptr(t, name="ring") for t in ring], int(ring[0].shape[0]),
        dev_ptr(state, torch.int64, "state"), dev_ptr(mask, torch.uint8, "mask"),
        dev_ptr(epoch_reward, torch.float64, "epoch_reward"), dev_ptr(ep_count, torch.int32, "ep_count"),
        dev_ptr(ep_log, name="ep_log"), int(ep_log.shape[0]), 0, N, D, A, int(bool(tanh_action)), stream_ptr(head.device)),
        "trl_synth_collect_step_f32")


def synth_reset(cur_obs, t_env, cur_step, episode_idx, ep_return, mask, seed_base):
    N, D = int(cur_obs.shape[0]), int(cur_obs.shape[1])
    check(lib().trl_synth_reset_f32(dev_ptr(cur_obs, name="cur_obs"), dev_ptr(t_env, torch.int32, "t_env"),
                                    dev_ptr(cur_step, torch.int32, "cur_step"),
                                    dev_ptr(episode_idx, torch.int32, "episode_idx"),
                                    dev_ptr(ep_return, name="ep_return"),
                                    dev_ptr(mask, torch.uint8, "mask", allow_none=True), N, D, int(seed_base),
                                    stream_ptr(cur_obs.device)), "trl_synth_reset_f32")


def gauss_logp(mean, acts, logstd, tanh_action, out=None):
    B, A = int(mean.shape[0]), int(mean.shape[1])
    if out is None:
        out = torch.empty((B,), dtype=torch.float32, device=mean.device)
    check(lib().trl_gauss_logp_f32(dev_ptr(mean, name="mean"), dev_ptr(acts, name="acts"),
                                   dev_ptr(logstd, name="logstd"), dev_ptr(out, name="out"), B, A,
                                   int(bool(tanh_action)), stream_ptr(mean.device)), "trl_gauss_logp_f32")
    return out


def linear_fwd(x, w, bias, act):
    """y = act(x @ w.T + bias); x (M, K), w (N, K) contiguous fp32 device tensors."""
    M, K = int(x.shape[0]), int(x.shape[1])
    N = int(w.shape[0])
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    need = lib().trl_linear_fwd_workspace(M, K, N)
    if need > 0:                                                         # few rows, long reduction: split-K + fold
        ws = torch.empty((need,), dtype=torch.float32, device=x.device)
        check(lib().trl_linear_fwd_splitk_f32(dev_ptr(x, name="x"), dev_ptr(w, name="w"),
                                              dev_ptr(bias, name="bias", allow_none=True), dev_ptr(y, name="y"),
                                              M, K, N, act, dev_ptr(ws, name="workspace"), stream_ptr(x.device)),
              "trl_linear_fwd_splitk_f32")
        return y
    check(lib().trl_linear_fwd_f32(dev_ptr(x, name="x"), dev_ptr(w, name="w"),
                                   dev_ptr(bias, name="bias", allow_none=True), dev_ptr(y, name="y"),
                                   M, K, N, act, stream_ptr(x.device)), "trl_linear_fwd_f32")
    return y


def ppo_generic_losses(mean, logstd, acts, advs, old_logp, v, rets, v_old, adv_raw, n_global, clip_para, entropy_coeff,
                       clipped_value_loss, tanh_action, loss_mode, d_logstd, info, workspace=None):
    """The loss half of a PPO / A2C minibatch for arbitrary network shapes; returns (d_mean (B, A), d_v (B, 1))."""
    B, A = int(mean.shape[0]), int(mean.shape[1])
    need = lib().trl_ppo_generic_losses_workspace(B, A)
    if need < 0:
        raise TrlError("ppo_generic_losses: unsupported sizes B=%d A=%d" % (B, A))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty((need,), dtype=torch.float64, device=mean.device)
    d_mean = torch.empty((B, A), dtype=torch.float32, device=mean.device)
    d_v = torch.empty((B, 1), dtype=torch.float32, device=mean.device)
    check(lib().trl_ppo_generic_losses_f32(
        dev_ptr(mean, name="mean"), dev_ptr(logstd, name="logstd"), dev_ptr(acts, name="acts"), dev_ptr(advs, name="advs"),
        dev_ptr(old_logp, name="old_logp", allow_none=True), dev_ptr(v, name="v"), dev_ptr(rets, name="rets"),
        dev_ptr(v_old, name="v_old", allow_none=True), dev_ptr(adv_raw, torch.float64, "adv_raw"), float(n_global), B, A,
        float(clip_para), float(entropy_coeff), int(bool(clipped_value_loss)), int(bool(tanh_action)), int(loss_mode),
        dev_ptr(d_mean, name="d_mean"), dev_ptr(d_v, name="d_v"), dev_ptr(d_logstd, name="d_logstd"),
        dev_ptr(info, torch.float64, "info"), dev_ptr(workspace, torch.float64, "workspace"), stream_ptr(mean.device)),
        "trl_ppo_generic_losses_f32")
    return d_mean, d_v


def adv_normalize(advs, adv_raw, n_global, eps=1e-5):
    out = torch.empty(int(advs.numel()), dtype=torch.float32, device=advs.device)
    check(lib().trl_adv_normalize_f32(dev_ptr(advs, name="advs"), dev_ptr(adv_raw, torch.float64, "adv_raw"), float(n_global),
                                      int(advs.numel()), float(eps), dev_ptr(out, name="out"), stream_ptr(advs.device)),
          "trl_adv_normalize_f32")
    return out


def trpo_surrogate(mean, logstd, acts, adv_n, tanh_action, entropy_coeff, d_logstd, info):
    n, A = int(mean.shape[0]), int(mean.shape[1])
    need = lib().trl_trpo_surrogate_workspace(n, A)
    if need < 0:
        raise TrlError("trpo_surrogate: unsupported sizes n=%d A=%d" % (n, A))
    ws = torch.empty((need,), dtype=torch.float64, device=mean.device)
    d_mean = torch.empty((n, A), dtype=torch.float32, device=mean.device)
    check(lib().trl_trpo_surrogate_f32(dev_ptr(mean, name="mean"), dev_ptr(logstd, name="logstd"), dev_ptr(acts, name="acts"),
                                       dev_ptr(adv_n, name="adv_n"), n, A, int(bool(tanh_action)), float(entropy_coeff),
                                       dev_ptr(d_mean, name="d_mean"), dev_ptr(d_logstd, name="d_logstd"),
                                       dev_ptr(info, torch.float64, "info"), dev_ptr(ws, torch.float64, "workspace"),
                                       stream_ptr(mean.device)), "trl_trpo_surrogate_f32")
    return d_mean


def jvp_gate(a, b, h, act):
    out = torch.empty_like(a)
    check(lib().trl_jvp_gate_f32(dev_ptr(a, name="a"), dev_ptr(b, name="b", allow_none=True), dev_ptr(h, name="h", allow_none=True),
                                 act, int(a.numel()), dev_ptr(out, name="out"), stream_ptr(a.device)), "trl_jvp_gate_f32")
    return out


def fisher_scale(d_mu, logstd):
    out = torch.empty_like(d_mu)
    check(lib().trl_fisher_scale_f32(dev_ptr(d_mu, name="d_mu"), dev_ptr(logstd, name="logstd"), int(d_mu.shape[0]),
                                     int(d_mu.shape[1]), dev_ptr(out, name="out"), stream_ptr(d_mu.device)), "trl_fisher_scale_f32")
    return out


def ratio_loss(logp_new, logp_old, adv_n, out):
    check(lib().trl_ratio_loss_f32(dev_ptr(logp_new, name="logp_new"), dev_ptr(logp_old, name="logp_old"),
                                   dev_ptr(adv_n, name="adv_n"), int(adv_n.numel()), dev_ptr(out, torch.float64, "out"),
                                   stream_ptr(adv_n.device)), "trl_ratio_loss_f32")
    return out


def mse_value_loss(v, rets, n_global, loss_sum):
    d_v = torch.empty((int(v.numel()), 1), dtype=torch.float32, device=v.device)
    check(lib().trl_mse_value_loss_f32(dev_ptr(v, name="v"), dev_ptr(rets, name="rets"), int(v.numel()), float(n_global),
                                       dev_ptr(d_v, name="d_v"), dev_ptr(loss_sum, torch.float64, "loss_sum"),
                                       stream_ptr(v.device)), "trl_mse_value_loss_f32")
    return d_v


def vmpo_losses(mean, target_mean, logstd, target_logstd, acts, adv_n, dual_state, tanh_action, eta_eps, alpha_eps, dual_lr,
                d_logstd, info):
    n, A = int(mean.shape[0]), int(mean.shape[1])
    need = lib().trl_vmpo_losses_workspace(n, A)
    if need < 0:
        raise TrlError("vmpo_losses: unsupported sizes n=%d A=%d" % (n, A))
    ws = torch.empty((need,), dtype=torch.float64, device=mean.device)
    d_mean = torch.empty((n, A), dtype=torch.float32, device=mean.device)
    check(lib().trl_vmpo_losses_f32(dev_ptr(mean, name="mean"), dev_ptr(target_mean, name="target_mean"),
                                    dev_ptr(logstd, name="logstd"), dev_ptr(target_logstd, name="target_logstd"),
                                    dev_ptr(acts, name="acts"), dev_ptr(adv_n, name="adv_n"), dev_ptr(dual_state, name="dual"),
                                    n, A, int(bool(tanh_action)), float(eta_eps), float(alpha_eps), float(dual_lr),
                                    dev_ptr(d_mean, name="d_mean"), dev_ptr(d_logstd, name="d_logstd"),
                                    dev_ptr(info, torch.float64, "info"), dev_ptr(ws, torch.float64, "workspace"),
                                    stream_ptr(mean.device)), "trl_vmpo_losses_f32")
    return d_mean


def _ptrs(tensors, name, allow_none=False):
    """Host array of device pointers for the grouped entry points (None -> NULL array when every entry is None)."""
    if tensors is None or all(t is None for t in tensors):
        if not allow_none:
            raise TrlError("%s: missing tensors" % name)
        return None
    return (C.c_void_p * len(tensors))(*[dev_ptr(t, name=name, allow_none=allow_none) for t in tensors])


def linear_fwd_group(xs, ws, biases, act):
    """[act(x_g @ w_g.T + b_g)] for G same-shaped layers in one launch (twin critics, targets, several inputs)."""
    G = len(xs)
    M, K, N = int(xs[0].shape[0]), int(xs[0].shape[1]), int(ws[0].shape[0])
    if any(tuple(x.shape) != (M, K) for x in xs) or any(tuple(w.shape) != (N, K) for w in ws):
        raise TrlError("linear_fwd_group: the layers of a group must have identical shapes")
    ys = [torch.empty((M, N), dtype=torch.float32, device=xs[0].device) for _ in range(G)]
    need = lib().trl_linear_fwd_workspace(M, K, N)
    if need > 0:                                                         # few rows, long reduction: split-K + fold, grouped
        wsp = torch.empty((G * need,), dtype=torch.float32, device=xs[0].device)
        check(lib().trl_linear_fwd_splitk_group_f32(G, _ptrs(xs, "x"), _ptrs(ws, "w"), _ptrs(biases, "bias", True),
                                                    _ptrs(ys, "y"), M, K, N, act, dev_ptr(wsp, name="workspace"),
                                                    stream_ptr(xs[0].device)), "trl_linear_fwd_splitk_group_f32")
        return ys
    check(lib().trl_linear_fwd_group_f32(G, _ptrs(xs, "x"), _ptrs(ws, "w"), _ptrs(biases, "bias", True), _ptrs(ys, "y"),
                                         M, K, N, act, stream_ptr(xs[0].device)), "trl_linear_fwd_group_f32")
    return ys


def outer_gate_group(dqs, ws, hs, act):
    """[dq_g w_g^T * act'(h_g)]: the gated input gradient of G one-output layers in one streaming launch."""
    M, N = int(hs[0].shape[0]), int(hs[0].shape[1])
    outs = [torch.empty((M, N), dtype=torch.float32, device=hs[0].device) for _ in hs]
    check(lib().trl_outer_gate_group_f32(len(hs), _ptrs(dqs, "dq"), _ptrs(ws, "w"), _ptrs(hs, "h"), _ptrs(outs, "out"),
                                         M, N, act, stream_ptr(hs[0].device)), "trl_outer_gate_group_f32")
    return outs


def mlp3_forward_ok(D, H1, H2, O):
    return bool(lib().trl_mlp3_forward_ok(int(D), int(H1), int(H2), int(O)))


def mlp3_forward_group(layers_list, xs, act, last_act, keep):
    """G networks D -> 256 -> 256 -> O on G inputs in one launch; keep[g]: write h1 / h2 of network g (its backward pass
    needs them).  Returns [(h1 or None, h2 or None, y)]."""
    G = len(xs)
    M, D = int(xs[0].shape[0]), int(xs[0].shape[1])
    O = int(layers_list[0][2][0].shape[0])
    dev = xs[0].device
    f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    h1s = [f(M, 256) if keep[g] else None for g in range(G)]
    h2s = [f(M, 256) if keep[g] else None for g in range(G)]
    ys = [f(M, O) for _ in range(G)]
    col = lambda k, j: _ptrs([ls[k][j] for ls in layers_list], "param", True)

    def opt_ptrs(ts):                                                   # per-entry NULLs allowed
        return (C.c_void_p * G)(*[dev_ptr(t, name="h", allow_none=True) for t in ts])
    check(lib().trl_mlp3_forward_group_f32(G, _ptrs(xs, "x"), col(0, 0), col(0, 1), col(1, 0), col(1, 1), col(2, 0), col(2, 1),
                                           opt_ptrs(h1s), opt_ptrs(h2s), _ptrs(ys, "y"), M, D, O, act, last_act,
                                           stream_ptr(dev)), "trl_mlp3_forward_group_f32")
    return list(zip(h1s, h2s, ys))


def linear_bwd_input_group(dys, y_gates, gate_act, ws):
    G = len(dys)
    M, N, K = int(dys[0].shape[0]), int(dys[0].shape[1]), int(ws[0].shape[1])
    dxs = [torch.empty((M, K), dtype=torch.float32, device=dys[0].device) for _ in range(G)]
    check(lib().trl_linear_bwd_input_group_f32(G, _ptrs(dys, "dy"), _ptrs(y_gates, "y_gate", True), gate_act,
                                               _ptrs(ws, "w"), _ptrs(dxs, "dx"), M, K, N, stream_ptr(dys[0].device)),
          "trl_linear_bwd_input_group_f32")
    return dxs


def linear_bwd_weight_group(dys, y_gates, gate_act, xs, dws, dbs, workspace=None):
    G = len(dys)
    M, N, K = int(dys[0].shape[0]), int(dys[0].shape[1]), int(xs[0].shape[1])
    need = G * lib().trl_linear_bwd_weight_workspace(M, K, N)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty((need,), dtype=torch.float32, device=dys[0].device)
    check(lib().trl_linear_bwd_weight_group_f32(G, _ptrs(dys, "dy"), _ptrs(y_gates, "y_gate", True), gate_act,
                                                _ptrs(xs, "x"), _ptrs(dws, "dw"), _ptrs(dbs, "db", True),
                                                dev_ptr(workspace, name="workspace"), M, K, N,
                                                stream_ptr(dys[0].device)), "trl_linear_bwd_weight_group_f32")


class FoldScope:
    """with FoldScope(device): the single-problem weight-gradient folds launched inside are recorded and run as ONE launch at
    exit (include/trl_hip.h trl_fold_scope_*).  Every layer inside needs its own workspace region."""

    def __init__(self, device):
        self.device = device

    def __enter__(self):
        check(lib().trl_fold_scope_begin(), "trl_fold_scope_begin")
        return self

    def __exit__(self, exc_type, exc, tb):
        rc = lib().trl_fold_scope_end(stream_ptr(self.device))       # always closes the scope, also on an error inside
        if exc_type is None:
            check(rc, "trl_fold_scope_end")
        return False


class FoldPlan:
    """Weight-gradient folds deferred to ONE launch at the end of a backward pass: `linear_bwd_weight_group(...,
    plan=)` runs only the split GEMM, leaving its partials in a slice of the plan's workspace, and `run()` folds every
    recorded (partials -> dW / db view) pair with trl_fold_partials_multi_f32."""

    def __init__(self, workspace, defer_gemm=False):
        self.ws, self.used, self.entries = workspace, 0, []
        # defer_gemm: the split GEMMs themselves wait too and run as ONE launch over all recorded layers
        # (trl_linear_bwd_weight_partials_multi_f32) in front of the fold
        self.defer_gemm, self.problems = defer_gemm, []

    def take(self, n):
        if self.used + n > self.ws.numel():
            raise TrlError("FoldPlan: workspace of %d floats is too small (need %d more)" % (self.ws.numel(), n))
        out = self.ws[self.used:self.used + n]
        self.used += n
        return out

    def _run_gemms(self):
        probs, self.problems = self.problems, []
        # Layers narrower than a 64 x 64 tile (first / last layers: 17, 23 inputs, 1, 12 outputs) go in their own launch:
        # mixed with the 256 x 256 ones the common grid is mostly empty workgroups and the launch takes 62 us at SAC's
        # shapes, the two separate ones 30 + 20 us.
        wide = [p for p in probs if p[5] >= 64 and p[6] >= 64]
        groups = [g for g in (wide, [p for p in probs if not (p[5] >= 64 and p[6] >= 64)]) if g]
        chunks = [g[lo:lo + 12] for g in groups for lo in range(0, len(g), 12)]
        for chunk in chunks:
            g = len(chunk)
            act = {p[2] for p in chunk if p[1] is not None}
            if len(act) > 1:
                raise TrlError("FoldPlan: one launch gates with one activation")
            M = int(chunk[0][0].shape[0])
            dys, gates, xs, wss = (C.c_void_p * g)(), (C.c_void_p * g)(), (C.c_void_p * g)(), (C.c_void_p * g)()
            Ks, Ns = (C.c_int * g)(), (C.c_int * g)()
            for j, (dy, gate, _act, x, ws, K, N) in enumerate(chunk):
                if int(dy.shape[0]) != M:
                    raise TrlError("FoldPlan: layers of one launch share the batch size")
                dys[j], xs[j], wss[j] = dev_ptr(dy, name="dy"), dev_ptr(x, name="x"), dev_ptr(ws, name="workspace")
                gates[j] = dev_ptr(gate, name="y_gate", allow_none=True)
                Ks[j], Ns[j] = K, N
            check(lib().trl_linear_bwd_weight_partials_multi_f32(g, dys, gates, act.pop() if act else ACT_NONE, xs, Ks, Ns, 1,
                                                                 wss, M, stream_ptr(self.ws.device)),
                  "trl_linear_bwd_weight_partials_multi_f32")

    def run(self):
        if self.problems:
            self._run_gemms()
        k = len(self.entries)
        if k == 0:
            return
        for lo in range(0, k, 32):
            chunk = self.entries[lo:lo + 32]
            c = len(chunk)
            parts, outs, ns, sp = (C.c_void_p * c)(), (C.c_void_p * c)(), (C.c_int * c)(), (C.c_int * c)()
            for j, (part, out, n, splits) in enumerate(chunk):
                parts[j], outs[j], ns[j], sp[j] = dev_ptr(part, name="partials"), dev_ptr(out, name="grad view"), n, splits
            check(lib().trl_fold_partials_multi_f32(c, parts, outs, ns, sp, stream_ptr(self.ws.device)),
                  "trl_fold_partials_multi_f32")
        self.entries, self.used = [], 0

    def tiles(self, grads):
        """Do the recorded gradient views cover the flat buffer `grads` exactly once, without gaps?  (What run_fused
        needs; a layer without a bias, or a parameter no fold entry writes, does not -- the caller then takes `run()` and
        the separate clip / Adam / Polyak launches.)"""
        pos, base = 0, grads.data_ptr()
        for part, out, n, splits in sorted(self.entries, key=lambda t: t[1].data_ptr()):
            if out.data_ptr() != base + 4 * pos:
                return False
            pos += n
        return pos == grads.numel()

    def run_fused(self, args, grads, target, target_off, tau, workspace, file=None):
        """`run()` and the update's last two launches in one (include/trl_hip.h trl_fold_clip_adam_polyak_f32): the folds
        land in `grads` (the flat gradient buffer every recorded view lives in -- the entries must cover it), then clip,
        Adam (`args`, device-resident step state) and the Polyak step of `target` against params[target_off:]."""
        if self.problems:
            self._run_gemms()
        base, total = grads.data_ptr(), grads.numel()
        ents = sorted(self.entries, key=lambda t: t[1].data_ptr())
        c = len(ents)
        parts, ns, sp = (C.c_void_p * c)(), (C.c_int * c)(), (C.c_int * c)()
        pos = 0
        for j, (part, out, n, splits) in enumerate(ents):
            if out.data_ptr() != base + 4 * pos:
                raise TrlError("FoldPlan.run_fused: the recorded gradient views do not tile the flat gradient buffer")
            parts[j], ns[j], sp[j] = dev_ptr(part, name="partials"), n, splits
            pos += n
        if pos != total:
            raise TrlError("FoldPlan.run_fused: %d of %d gradients have a fold entry" % (pos, total))
        raw, ring = file if file is not None else (None, None)
        check(lib().trl_fold_clip_adam_polyak_f32(
            c, parts, ns, sp, C.byref(args), dev_ptr(target, name="target"), int(target_off), int(target.numel()), float(tau),
            dev_ptr(raw, torch.uint8, "raw", allow_none=True), 0 if raw is None else int(raw.numel()),
            dev_ptr(ring, torch.uint8, "ring", allow_none=True), 0 if ring is None else int(ring.shape[0]),
            dev_ptr(workspace, torch.uint8, "workspace"), stream_ptr(grads.device)), "trl_fold_clip_adam_polyak_f32")
        self.entries, self.used = [], 0


def linear_bwd_weight_partials_group(dys, y_gates, gate_act, xs, dws, dbs, plan):
    G = len(dys)
    M, N, K = int(dys[0].shape[0]), int(dys[0].shape[1]), int(xs[0].shape[1])
    if plan.defer_gemm and dbs is not None and dbs[0] is not None:
        splits = lib().trl_linear_bwd_weight_multi_splits(M, K, N)
        for i in range(G):
            ws = plan.take(splits * (N * K + N))
            gate = y_gates[i] if (y_gates is not None and gate_act != ACT_NONE) else None
            plan.problems.append((dys[i], gate, gate_act, xs[i], ws, K, N))
            plan.entries.append((ws[:splits * N * K], dws[i], N * K, splits))
            plan.entries.append((ws[splits * N * K:], dbs[i], N, splits))
        return
    splits = lib().trl_linear_bwd_weight_splits(M, K, N)
    per = splits * (N * K + N)
    ws = plan.take(G * per)
    want_db = dbs is not None and dbs[0] is not None
    check(lib().trl_linear_bwd_weight_partials_group_f32(G, _ptrs(dys, "dy"), _ptrs(y_gates, "y_gate", True), gate_act,
                                                         _ptrs(xs, "x"), int(want_db), dev_ptr(ws, name="workspace"),
                                                         M, K, N, stream_ptr(dys[0].device)),
          "trl_linear_bwd_weight_partials_group_f32")
    for i in range(G):
        base = i * per
        plan.entries.append((ws[base:base + splits * N * K], dws[i], N * K, splits))
        if want_db:
            plan.entries.append((ws[base + splits * N * K:base + per], dbs[i], N, splits))


def linear_bwd_input(dy, y_gate, gate_act, w):
    M, N = int(dy.shape[0]), int(dy.shape[1])
    K = int(w.shape[1])
    dx = torch.empty((M, K), dtype=torch.float32, device=dy.device)
    need = lib().trl_linear_bwd_input_workspace(M, K, N)
    if need > 0:                                   # few output tiles behind a long reduction: split over slices + fold
        ws = torch.empty((need,), dtype=torch.float32, device=dy.device)
        check(lib().trl_linear_bwd_input_splitk_f32(dev_ptr(dy, name="dy"), dev_ptr(y_gate, name="y_gate", allow_none=True),
                                                    gate_act, dev_ptr(w, name="w"), dev_ptr(dx, name="dx"),
                                                    dev_ptr(ws, name="workspace"), M, K, N, stream_ptr(dy.device)),
              "trl_linear_bwd_input_splitk_f32")
        return dx
    check(lib().trl_linear_bwd_input_f32(dev_ptr(dy, name="dy"), dev_ptr(y_gate, name="y_gate", allow_none=True),
                                         gate_act, dev_ptr(w, name="w"), dev_ptr(dx, name="dx"), M, K, N,
                                         stream_ptr(dy.device)), "trl_linear_bwd_input_f32")
    return dx


def linear_bwd_weight(dy, y_gate, gate_act, x, dw=None, db=None, need_bias=True, workspace=None):
    """dw (N, K) and db (N) of a dense layer; `dw` / `db` may be views into a flat gradient buffer."""
    M, N = int(dy.shape[0]), int(dy.shape[1])
    K = int(x.shape[1])
    if dw is None:
        dw = torch.empty((N, K), dtype=torch.float32, device=dy.device)
    if db is None and need_bias:
        db = torch.empty((N,), dtype=torch.float32, device=dy.device)
    need = lib().trl_linear_bwd_weight_workspace(M, K, N)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty((need,), dtype=torch.float32, device=dy.device)
    check(lib().trl_linear_bwd_weight_f32(dev_ptr(dy, name="dy"), dev_ptr(y_gate, name="y_gate", allow_none=True),
                                          gate_act, dev_ptr(x, name="x"), dev_ptr(dw, name="dw"),
                                          dev_ptr(db, name="db", allow_none=True), dev_ptr(workspace, name="workspace"),
                                          M, K, N, stream_ptr(dy.device)), "trl_linear_bwd_weight_f32")
    return dw, db


def concat2(a, b):
    rows, fa, fb = int(a.shape[0]), int(a.shape[1]), int(b.shape[1])
    out = torch.empty((rows, fa + fb), dtype=torch.float32, device=a.device)
    check(lib().trl_concat2_f32(dev_ptr(a, name="a"), dev_ptr(b, name="b"), dev_ptr(out, name="out"), rows, fa, fb,
                                stream_ptr(a.device)), "trl_concat2_f32")
    return out


def rsample_fwd(head, eps, tanh_action=True):
    B, A = int(eps.shape[0]), int(eps.shape[1])
    act = torch.empty((B, A), dtype=torch.float32, device=head.device)
    logp = torch.empty((B,), dtype=torch.float32, device=head.device)
    check(lib().trl_tanh_gauss_rsample_fwd_f32(dev_ptr(head, name="head"), dev_ptr(eps, name="eps"),
                                               dev_ptr(act, name="act"), dev_ptr(logp, name="logp"), B, A,
                                               int(bool(tanh_action)), stream_ptr(head.device)),
          "trl_tanh_gauss_rsample_fwd_f32")
    return act, logp


def rsample_bwd(head, eps, act, d_act, d_logp_ptr, d_logp_mul, w_std, w_mean, tanh_action=True):
    B, A = int(eps.shape[0]), int(eps.shape[1])
    d_head = torch.empty((B, 2 * A), dtype=torch.float32, device=head.device)
    check(lib().trl_tanh_gauss_rsample_bwd_f32(dev_ptr(head, name="head"), dev_ptr(eps, name="eps"),
                                               dev_ptr(act, name="act"), dev_ptr(d_act, name="d_act"),
                                               dev_ptr(d_logp_ptr, name="d_logp_ptr", allow_none=True),
                                               float(d_logp_mul), float(w_std), float(w_mean),
                                               dev_ptr(d_head, name="d_head"), B, A, int(bool(tanh_action)),
                                               stream_ptr(head.device)), "trl_tanh_gauss_rsample_bwd_f32")
    return d_head


def rsample_bwd_cols(head, eps, act, dx1, dx2, off, d_logp_ptr, d_logp_mul, w_std, w_mean, tanh_action=True):
    """rsample_bwd with d_act = dx1[:, off:off+A] + dx2[:, off:off+A] read in place (dx2 may be None)."""
    B, A = int(eps.shape[0]), int(eps.shape[1])
    d_head = torch.empty((B, 2 * A), dtype=torch.float32, device=head.device)
    check(lib().trl_tanh_gauss_rsample_bwd_cols_f32(
        dev_ptr(head, name="head"), dev_ptr(eps, name="eps"), dev_ptr(act, name="act"), dev_ptr(dx1, name="dx1"),
        dev_ptr(dx2, name="dx2", allow_none=True), int(dx1.shape[1]), int(off),
        dev_ptr(d_logp_ptr, name="d_logp_ptr", allow_none=True), float(d_logp_mul), float(w_std), float(w_mean),
        dev_ptr(d_head, name="d_head"), B, A, int(bool(tanh_action)), stream_ptr(head.device)),
        "trl_tanh_gauss_rsample_bwd_cols_f32")
    return d_head


def sac_policy_grad_ok(dys, ws, A):
    H = int(dys[0].shape[1])
    return len(dys) in (1, 2) and bool(lib().trl_sac_policy_grad_supported(H, int(A))) and \
        all(d.data_ptr() % 16 == 0 and d.is_contiguous() and d.dtype == torch.float32 for d in dys)


def sac_policy_grad(head, eps, act, dys, ys, gate_act, ws, off, d_logp_ptr, d_logp_mul, w_std, w_mean, tanh_action=True,
                    head_layer=None):
    """d_head of the policy loss from the gradients `dys` at the critics' first hidden layer (outputs `ys`, weights `ws`
    (H, D + A)): the action columns [off, off + A) of that layer's input gradient, summed over the critics, pushed through
    the sampler's backward -- one launch (include/trl_hip.h trl_sac_policy_grad_f32).
    head_layer = (W3 (2A, H), H2 (B, H), act): also returns dZ2 = (d_head W3) * act'(H2), the gradient at the policy's
    second hidden layer already gated for the layer below (None when the shapes do not fit: H2's width must be H)."""
    B, A, H = int(eps.shape[0]), int(eps.shape[1]), int(dys[0].shape[1])
    d_head = torch.empty((B, 2 * A), dtype=torch.float32, device=head.device)
    gated = gate_act != ACT_NONE and ys is not None and all(y is not None for y in ys)
    hw = hh = hdz = None
    hact = ACT_NONE
    if head_layer is not None:
        w3, h2, hact = head_layer
        if tuple(w3.shape) == (2 * A, H) and tuple(h2.shape) == (B, H) and w3.is_contiguous() and h2.is_contiguous() \
                and w3.data_ptr() % 16 == 0 and h2.data_ptr() % 16 == 0:
            hw, hh, hdz = w3, h2, torch.empty((B, H), dtype=torch.float32, device=head.device)
    check(lib().trl_sac_policy_grad_f32(
        len(dys), _ptrs(dys, "dy"), _ptrs(ys, "y") if gated else None, gate_act if gated else ACT_NONE, _ptrs(ws, "w"), H,
        int(ws[0].shape[1]), int(off), dev_ptr(head, name="head"), dev_ptr(eps, name="eps"), dev_ptr(act, name="act"),
        dev_ptr(d_logp_ptr, name="d_logp_ptr", allow_none=True), float(d_logp_mul), float(w_std), float(w_mean),
        dev_ptr(d_head, name="d_head"), B, A, int(bool(tanh_action)), dev_ptr(hw, name="head_w", allow_none=True),
        dev_ptr(hh, name="head_h", allow_none=True), int(hact), dev_ptr(hdz, name="head_dz", allow_none=True),
        stream_ptr(head.device)), "trl_sac_policy_grad_f32")
    return d_head if head_layer is None else (d_head, hdz)


def sac_samples(head, head2, eps1, eps2, obs, acts, next_obs, tanh_action=True, philox=None, mom_part=None):
    """(new_a, logp, next_a, next_logp, x_sa, x_next, x_new) of one SAC update in one launch.  philox = (step_state,
    seed): the two noise draws are made inside the launch from the device-resident update count; eps1 then RECEIVES the
    first draw (eps2 is not touched).  mom_part (ceil(B / 64), 12) float64: also receives the per-wave partial moments
    of the clamped log_std / log_prob / mean that `sac_losses(..., fold=)` turns into the logged statistics."""
    B, A, D = int(eps1.shape[0]), int(eps1.shape[1]), int(obs.shape[1])
    f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=head.device)
    new_a, logp, next_a, next_logp = f(B, A), f(B), f(B, A), f(B)
    x_sa, x_next, x_new = f(B, D + A), f(B, D + A), f(B, D + A)
    outs = [dev_ptr(t, name="out") for t in (new_a, logp, next_a, next_logp, x_sa, x_next, x_new)]
    if mom_part is not None and mom_part.numel() < 12 * ((B + 63) // 64):
        raise TrlError("sac_samples: mom_part holds fewer than ceil(B / 64) rows of 12")
    state, seed = philox if philox is not None else (None, 0)
    check(lib().trl_sac_samples_f32(dev_ptr(head, name="head"), dev_ptr(head2, name="head2"),
                                    dev_ptr(eps1, name="eps1"), dev_ptr(eps2, name="eps2", allow_none=philox is not None),
                                    dev_ptr(state, torch.float64, "step_state", allow_none=True), int(seed),
                                    dev_ptr(obs, name="obs"), dev_ptr(acts, name="acts"),
                                    dev_ptr(next_obs, name="next_obs"), *outs, B, D, A, int(bool(tanh_action)),
                                    dev_ptr(mom_part, torch.float64, "mom_part", allow_none=True), stream_ptr(head.device)),
          "trl_sac_samples_f32")
    return new_a, logp, next_a, next_logp, x_sa, x_next, x_new


def sac_alpha_step(logp, target_entropy, lr, state, out, beta1=0.9, beta2=0.999, eps=1e-8):
    check(lib().trl_sac_alpha_step_f32(dev_ptr(logp, name="logp"), int(logp.numel()), float(target_entropy),
                                       float(lr), beta1, beta2, eps, dev_ptr(state, name="state"),
                                       dev_ptr(out, name="out"), stream_ptr(logp.device)), "trl_sac_alpha_step_f32")


def sac_losses(q1, q2, tq1, tq2, logp_next, rew, term, q1n, q2n, logp, alpha, gamma, sums, alpha_step=None, fold=None):
    """alpha_step = (target_entropy, lr, state, out): the temperature step is taken inside the launch (then `alpha` may be
    None); fold = (mom_part, A, mom_out12): sac_samples' partial moments are folded into the logged statistics."""
    B = int(q1.numel())
    outs = [torch.empty((B, 1), dtype=torch.float32, device=q1.device) for _ in range(4)]
    ent, lr, state, aout = alpha_step if alpha_step is not None else (0.0, 0.0, None, None)
    part, A, mom_out = fold if fold is not None else (None, 0, None)
    ins = [q1, q2, tq1, tq2, logp_next, rew, term, q1n, q2n, logp]
    check(lib().trl_sac_losses_f32(*[dev_ptr(t, name="in%d" % i) for i, t in enumerate(ins)],
                                   dev_ptr(alpha, name="alpha", allow_none=alpha_step is not None), float(gamma), B,
                                   *[dev_ptr(t, name="out") for t in outs], dev_ptr(sums, torch.float64, "sums"),
                                   dev_ptr(state, name="alpha_state", allow_none=True),
                                   dev_ptr(aout, name="alpha_out", allow_none=True), float(ent), float(lr),
                                   0.9, 0.999, 1e-8, dev_ptr(part, torch.float64, "mom_part", allow_none=True), int(A),
                                   dev_ptr(mom_out, torch.float64, "mom_out", allow_none=True),
                                   stream_ptr(q1.device)), "trl_sac_losses_f32")
    return outs


def slice_add(x1, x2, off, A):
    rows, ld = int(x1.shape[0]), int(x1.shape[1])
    out = torch.empty((rows, A), dtype=torch.float32, device=x1.device)
    check(lib().trl_slice_add_f32(dev_ptr(x1, name="x1"), dev_ptr(x2, name="x2", allow_none=True), dev_ptr(out, name="out"), rows, ld,
                                  off, A, stream_ptr(x1.device)), "trl_slice_add_f32")
    return out


def polyak(target, source, tau):
    check(lib().trl_polyak_f32(dev_ptr(target, name="target"), dev_ptr(source, name="source"), int(target.numel()),
                               float(tau), stream_ptr(target.device)), "trl_polyak_f32")


def moments(x, out4, ld=None, off=0, width=None, lo=float("-inf"), hi=float("inf")):
    moments_multi([(x, out4, ld, off, width, lo, hi)])


def moments_multi(specs, ring=None):
    """Several `moments` in one launch; specs: up to 4 of (x, out4, ld, off, width, lo, hi).
    ring = (raw uint8 block holding every out4, ring (slots, raw bytes) uint8, device float64 update counter): the launch
    also files `raw` into ring[(counter - 1) % slots]."""
    k = len(specs)
    xs, outs = (C.c_void_p * k)(), (C.c_void_p * k)()
    ns, lds, offs, ws = (C.c_int64 * k)(), (C.c_int * k)(), (C.c_int * k)(), (C.c_int * k)()
    los, his = (C.c_float * k)(), (C.c_float * k)()
    for j, (x, out4, ld, off, width, lo, hi) in enumerate(specs):
        ld = int(x.shape[-1]) if ld is None else ld
        xs[j], outs[j] = dev_ptr(x, name="x"), dev_ptr(out4, torch.float64, "out4")
        ns[j], lds[j], offs[j], ws[j] = int(x.numel()), ld, off, (ld - off if width is None else width)
        los[j], his[j] = lo, hi
    raw, slots, counter = ring if ring is not None else (None, None, None)
    if slots is not None and (slots.dim() != 2 or int(slots.shape[1]) != raw.numel() or not slots.is_contiguous()):
        raise TrlError("moments_multi: ring rows must be statistics blocks")
    check(lib().trl_moments_multi_f64(k, xs, ns, lds, offs, ws, los, his, outs,
                                      dev_ptr(raw, torch.uint8, "raw", allow_none=True), 0 if raw is None else int(raw.numel()),
                                      dev_ptr(slots, torch.uint8, "ring", allow_none=True),
                                      0 if slots is None else int(slots.shape[0]),
                                      dev_ptr(counter, torch.float64, "counter", allow_none=True),
                                      stream_ptr(specs[0][0].device)), "trl_moments_multi_f64")


def philox_normal(out, seed, counter):
    check(lib().trl_philox_normal_f32(dev_ptr(out, name="out"), int(out.numel()), int(seed), int(counter),
                                      stream_ptr(out.device)), "trl_philox_normal_f32")
    return out


def synth_env_step(cur_obs, act, env_A, env_B, t_env, reward_scale, horizon, next_obs, rewards, dones):
    N, D, A = int(cur_obs.shape[0]), int(cur_obs.shape[1]), int(act.shape[1])
    check(lib().trl_synth_env_step_f32(dev_ptr(cur_obs, name="cur_obs"), dev_ptr(act, name="act"),
                                       dev_ptr(env_A, name="env_A"), dev_ptr(env_B, name="env_B"),
                                       dev_ptr(t_env, torch.int32, "t_env"), float(reward_scale), int(horizon),
                                       dev_ptr(next_obs, name="next_obs"), dev_ptr(rewards, name="rewards"),
                                       dev_ptr(dones, name="dones"), N, D, A, stream_ptr(cur_obs.device)),
          "trl_synth_env_step_f32")


def collector_bookkeep(rewards, dones, cur_step, ep_return, max_frames, mask, epoch_reward, ep_count, ep_log, step):
    N = int(rewards.numel())
    check(lib().trl_collector_bookkeep_f32(dev_ptr(rewards, name="rewards"), dev_ptr(dones, name="dones"),
                                           dev_ptr(cur_step, torch.int32, "cur_step"),
                                           dev_ptr(ep_return, name="ep_return"), int(max_frames),
                                           dev_ptr(mask, torch.uint8, "mask"),
                                           dev_ptr(epoch_reward, torch.float64, "epoch_reward"),
                                           dev_ptr(ep_count, torch.int32, "ep_count"), dev_ptr(ep_log, name="ep_log"),
                                           int(ep_log.shape[0]), int(step), N, stream_ptr(rewards.device)),
          "trl_collector_bookkeep_f32")


def im2col(x, kh, kw, sh, sw, scale=None, shift=0.0):
    """x: (B, H, W, C) fp32 channels-last, or (B, C, H, W) uint8 when `scale` is given."""
    if x.dtype == torch.uint8:
        B, Cc, H, W = (int(v) for v in x.shape)
    else:
        B, H, W, Cc = (int(v) for v in x.shape)
    Ho, Wo = (H - kh) // sh + 1, (W - kw) // sw + 1
    cols = torch.empty((B * Ho * Wo, Cc * kh * kw), dtype=torch.float32, device=x.device)
    if x.dtype == torch.uint8:
        check(lib().trl_im2col_u8_nchw(dev_ptr(x, torch.uint8, "x"), dev_ptr(cols, name="cols"), B, Cc, H, W, kh, kw,
                                       sh, sw, float(1.0 if scale is None else scale), float(shift),
                                       stream_ptr(x.device)), "trl_im2col_u8_nchw")
    else:
        check(lib().trl_im2col_f32(dev_ptr(x, name="x"), dev_ptr(cols, name="cols"), B, Cc, H, W, kh, kw, sh, sw,
                                   stream_ptr(x.device)), "trl_im2col_f32")
    return cols, (B, Ho, Wo)


def conv_u8_implicit_ok(frames, kh, kw, sh, sw):
    """Geometry the implicit-GEMM first-layer kernels accept (include/trl_hip.h K16b)."""
    return kw % 4 == 0 and sw % 4 == 0 and int(frames.shape[3]) % 4 == 0


def conv_fwd_u8(frames, w, bias, kh, kw, sh, sw, scale, shift, act, perm=None, dx=None):
    """act(conv2d(frames * scale + shift, w) + bias) on (B, C, H, W) uint8 frames; returns ((B*Ho*Wo, Cout), (B, Ho, Wo)).
    Riders of the launch (include/trl_hip.h trl_conv_riders_t; then a third return value (perm outs, dx workspaces)):
    perm: [(weight (Cout, C*kh*kw) of a LATER conv layer, its C, its kh*kw)] (<= 4) -> those weights re-ordered to the
    (i, j, c) reduction order `conv_fwd_nhwc(..., w_perm=True)` reads;  dx: [(weight, Cin, kh, kw, sh, sw)] (<= 4) -> the
    workspaces `conv_bwd_input_nhwc(..., prep=)` takes."""
    B, Cc, H, W = (int(v) for v in frames.shape)
    Ho, Wo = (H - kh) // sh + 1, (W - kw) // sw + 1
    Cout = int(w.shape[0])
    y = torch.empty((B * Ho * Wo, Cout), dtype=torch.float32, device=frames.device)
    jobs, dxs = list(perm or []), list(dx or [])
    outs = [torch.empty_like(wl, memory_format=torch.contiguous_format) for wl, _, _ in jobs]
    wss = [torch.empty((lib().trl_conv_bwd_input_nhwc_workspace(cin, int(wl.shape[0]), a, b),), dtype=torch.float32, device=frames.device)
           for wl, cin, a, b, _, _ in dxs]
    r = None
    if jobs or dxs:
        r = ConvRiders()
        r.n_perm, r.n_dx = len(jobs), len(dxs)
        for k, ((wl, c, khw), o) in enumerate(zip(jobs, outs)):
            r.perm_src[k], r.perm_dst[k] = dev_ptr(wl, name="perm weight"), dev_ptr(o, name="perm out")
            r.perm_cout[k], r.perm_c[k], r.perm_khw[k] = int(wl.shape[0]), int(c), int(khw)
        for k, ((wl, cin, a, b, c, d), ws) in enumerate(zip(dxs, wss)):
            r.dx_w[k], r.dx_ws[k] = dev_ptr(wl, name="dx weight"), dev_ptr(ws, name="dx workspace")
            r.dx_cin[k], r.dx_cout[k], r.dx_kh[k], r.dx_kw[k], r.dx_sh[k], r.dx_sw[k] = int(cin), int(wl.shape[0]), a, b, c, d
    check(lib().trl_conv_fwd_u8_f32(dev_ptr(frames, torch.uint8, "frames"), dev_ptr(w, name="w"),
                                    dev_ptr(bias, name="bias", allow_none=True), dev_ptr(y, name="y"), B, Cc, H, W,
                                    kh, kw, sh, sw, float(scale), float(shift), Cout, act,
                                    C.byref(r) if r is not None else None, stream_ptr(frames.device)),
          "trl_conv_fwd_u8_f32")
    if perm is None and dx is None:
        return y, (B, Ho, Wo)
    return y, (B, Ho, Wo), (outs, wss)


def conv_fwd_u8_pair(frames_a, w_a, bias_a, frames_b, w_b, bias_b, kh, kw, sh, sw, scale, shift, act, perm=None, dx=None):
    """`conv_fwd_u8` of two same-geometry problems (two networks of one architecture on two frame batches of one shape) as ONE
    launch; the riders (`perm` / `dx`, together at most 4 of either kind) may come from both networks.  Returns
    (y_a, y_b, (B, Ho, Wo), (perm outs, dx workspaces))."""
    if tuple(frames_a.shape) != tuple(frames_b.shape) or tuple(w_a.shape) != tuple(w_b.shape):
        raise TrlError("conv_fwd_u8_pair: the two problems must have one geometry")
    B, Cc, H, W = (int(v) for v in frames_a.shape)
    Ho, Wo = (H - kh) // sh + 1, (W - kw) // sw + 1
    Cout = int(w_a.shape[0])
    ya = torch.empty((B * Ho * Wo, Cout), dtype=torch.float32, device=frames_a.device)
    yb = torch.empty_like(ya)
    jobs, dxs = list(perm or []), list(dx or [])
    if len(jobs) > 4 or len(dxs) > 4:
        raise TrlError("conv_fwd_u8_pair: at most 4 riders of either kind")
    outs = [torch.empty_like(wl, memory_format=torch.contiguous_format) for wl, _, _ in jobs]
    wss = [torch.empty((lib().trl_conv_bwd_input_nhwc_workspace(cin, int(wl.shape[0]), a, b),), dtype=torch.float32, device=ya.device)
           for wl, cin, a, b, _, _ in dxs]
    r = None
    if jobs or dxs:
        r = ConvRiders()
        r.n_perm, r.n_dx = len(jobs), len(dxs)
        for k, ((wl, c, khw), o) in enumerate(zip(jobs, outs)):
            r.perm_src[k], r.perm_dst[k] = dev_ptr(wl, name="perm weight"), dev_ptr(o, name="perm out")
            r.perm_cout[k], r.perm_c[k], r.perm_khw[k] = int(wl.shape[0]), int(c), int(khw)
        for k, ((wl, cin, a, b, c, d), ws) in enumerate(zip(dxs, wss)):
            r.dx_w[k], r.dx_ws[k] = dev_ptr(wl, name="dx weight"), dev_ptr(ws, name="dx workspace")
            r.dx_cin[k], r.dx_cout[k], r.dx_kh[k], r.dx_kw[k], r.dx_sh[k], r.dx_sw[k] = int(cin), int(wl.shape[0]), a, b, c, d
    check(lib().trl_conv_fwd_u8_pair_f32(dev_ptr(frames_a, torch.uint8, "frames_a"), dev_ptr(w_a, name="w_a"),
                                         dev_ptr(bias_a, name="bias_a", allow_none=True), dev_ptr(ya, name="y_a"),
                                         dev_ptr(frames_b, torch.uint8, "frames_b"), dev_ptr(w_b, name="w_b"),
                                         dev_ptr(bias_b, name="bias_b", allow_none=True), dev_ptr(yb, name="y_b"), B, Cc, H, W,
                                         kh, kw, sh, sw, float(scale), float(shift), Cout, act,
                                         C.byref(r) if r is not None else None, stream_ptr(ya.device)),
          "trl_conv_fwd_u8_pair_f32")
    return ya, yb, (B, Ho, Wo), (outs, wss)


def conv_bwd_weight_u8(dy, y_gate, gate_act, frames, kh, kw, sh, sw, scale, shift, dw, db, workspace=None):
    B, Cc, H, W = (int(v) for v in frames.shape)
    Cout = int(dy.shape[1])
    need = lib().trl_conv_bwd_weight_workspace(B, Cc, H, W, kh, kw, sh, sw, Cout)
    if need < 0:
        raise TrlError("conv_bwd_weight_u8: bad geometry")
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty((need,), dtype=torch.float32, device=dy.device)
    check(lib().trl_conv_bwd_weight_u8_f32(dev_ptr(dy, name="dy"), dev_ptr(y_gate, name="y_gate", allow_none=True), gate_act,
                                           dev_ptr(frames, torch.uint8, "frames"), dev_ptr(dw, name="dw"),
                                           dev_ptr(db, name="db", allow_none=True), dev_ptr(workspace, name="workspace"),
                                           B, Cc, H, W, kh, kw, sh, sw, float(scale), float(shift), Cout,
                                           stream_ptr(dy.device)), "trl_conv_bwd_weight_u8_f32")
    return dw, db


def conv_fwd_nhwc(x, w, bias, kh, kw, sh, sw, act, out_chw=False, w_perm=False):
    """act(conv2d(x) + bias) on (B, H, W, C) fp32 channels-last activations, C % 4 == 0; w (Cout, C*kh*kw) is the
    nn.Conv2d weight as stored.  Returns ((B*Ho*Wo, Cout), (B, Ho, Wo)); with `out_chw` the result is stored
    (B, Cout, Ho*Wo) -- nn.Flatten's order -- and returned as (B, Cout*Ho*Wo).  w_perm: `w` is already in the (i, j, c)
    reduction order (conv_fwd_u8(perm=))."""
    B, H, W, Cc = (int(v) for v in x.shape)
    Ho, Wo = (H - kh) // sh + 1, (W - kw) // sw + 1
    Cout = int(w.shape[0])
    y = torch.empty((B, Cout * Ho * Wo) if out_chw else (B * Ho * Wo, Cout), dtype=torch.float32, device=x.device)
    check(lib().trl_conv_fwd_nhwc_f32(dev_ptr(x, name="x"), dev_ptr(w, name="w"), dev_ptr(bias, name="bias", allow_none=True),
                                      dev_ptr(y, name="y"), B, Cc, H, W, kh, kw, sh, sw, Cout, act, int(bool(out_chw)),
                                      int(bool(w_perm)), stream_ptr(x.device)), "trl_conv_fwd_nhwc_f32")
    return y, (B, Ho, Wo)


def conv_fwd_nhwc_group(xs, ws, biases, kh, kw, sh, sw, act, out_chw=False, w_perm=False):
    """`conv_fwd_nhwc` of G same-geometry layers (different inputs / weights) in one launch; returns ([y_g], (B, Ho, Wo))."""
    B, H, W, Cc = (int(v) for v in xs[0].shape)
    if any(tuple(x.shape) != tuple(xs[0].shape) for x in xs) or any(tuple(w.shape) != tuple(ws[0].shape) for w in ws):
        raise TrlError("conv_fwd_nhwc_group: the layers of a group share one geometry")
    Ho, Wo = (H - kh) // sh + 1, (W - kw) // sw + 1
    Cout = int(ws[0].shape[0])
    ys = [torch.empty((B, Cout * Ho * Wo) if out_chw else (B * Ho * Wo, Cout), dtype=torch.float32, device=xs[0].device)
          for _ in xs]
    check(lib().trl_conv_fwd_nhwc_group_f32(len(xs), _ptrs(xs, "x"), _ptrs(ws, "w"), _ptrs(biases, "bias", True),
                                            _ptrs(ys, "y"), B, Cc, H, W, kh, kw, sh, sw, Cout, act, int(bool(out_chw)),
                                            int(bool(w_perm)), stream_ptr(xs[0].device)), "trl_conv_fwd_nhwc_group_f32")
    return ys, (B, Ho, Wo)


def conv_bwd_weight_nhwc(dy, y_gate, gate_act, x, kh, kw, sh, sw, dw, db, workspace=None):
    B, H, W, Cc = (int(v) for v in x.shape)
    Cout = int(dy.shape[1])
    need = lib().trl_conv_bwd_weight_workspace(B, Cc, H, W, kh, kw, sh, sw, Cout)
    if need < 0:
        raise TrlError("conv_bwd_weight_nhwc: bad geometry")
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty((need,), dtype=torch.float32, device=dy.device)
    check(lib().trl_conv_bwd_weight_nhwc_f32(dev_ptr(dy, name="dy"), dev_ptr(y_gate, name="y_gate", allow_none=True), gate_act,
                                             dev_ptr(x, name="x"), dev_ptr(dw, name="dw"),
                                             dev_ptr(db, name="db", allow_none=True), dev_ptr(workspace, name="workspace"),
                                             B, Cc, H, W, kh, kw, sh, sw, Cout, stream_ptr(dy.device)),
          "trl_conv_bwd_weight_nhwc_f32")
    return dw, db


def col2im(dcols, B, Cc, H, W, kh, kw, sh, sw):
    dx = torch.empty((B, H, W, Cc), dtype=torch.float32, device=dcols.device)
    check(lib().trl_col2im_f32(dev_ptr(dcols, name="dcols"), dev_ptr(dx, name="dx"), B, Cc, H, W, kh, kw, sh, sw,
                               stream_ptr(dcols.device)), "trl_col2im_f32")
    return dx


def conv_bwd_input_ok(Cin, Cout, kh, kw, sh, sw):
    return bool(lib().trl_conv_bwd_input_nhwc_ok(int(Cin), int(Cout), kh, kw, sh, sw))


def conv_bwd_input_prep(layers, device):
    """layers: [(weight (Cout, Cin*kh*kw), Cin, kh, kw, sh, sw)] -> [workspace holding the layer's re-ordered weights], ONE
    launch for all of them (pass each as `prep=` to conv_bwd_input_nhwc)."""
    n = len(layers)
    wss = [torch.empty((lib().trl_conv_bwd_input_nhwc_workspace(cin, int(w.shape[0]), kh, kw),), dtype=torch.float32, device=device)
           for w, cin, kh, kw, sh, sw in layers]
    ints = lambda vals: (C.c_int * n)(*[int(v) for v in vals])
    check(lib().trl_conv_bwd_input_nhwc_prep_f32(
        n, _ptrs([w for w, *_ in layers], "weight"), _ptrs(wss, "workspace"), ints(l[1] for l in layers),
        ints(int(l[0].shape[0]) for l in layers), ints(l[2] for l in layers), ints(l[3] for l in layers),
        ints(l[4] for l in layers), ints(l[5] for l in layers), stream_ptr(device)), "trl_conv_bwd_input_nhwc_prep_f32")
    return wss


def conv_bwd_input_nhwc(dy, y_gate, gate_act, weight, B, Cin, H, W, kh, kw, sh, sw, x_gate=None, x_gate_act=ACT_NONE, prep=None):
    """dx (B, H, W, Cin) of a conv layer from dy (B*Ho*Wo, Cout), its activation output and the (Cout, Cin*kh*kw) weight;
    with x_gate (the layer's INPUT activations) the result is already multiplied by act'(x_gate).  prep: this layer's
    workspace from conv_bwd_input_prep (its weights are re-ordered already)."""
    Cout = int(weight.shape[0])
    dx = torch.empty((B, H, W, Cin), dtype=torch.float32, device=dy.device)
    ws = prep if prep is not None else \
        torch.empty((lib().trl_conv_bwd_input_nhwc_workspace(Cin, Cout, kh, kw),), dtype=torch.float32, device=dy.device)
    check(lib().trl_conv_bwd_input_nhwc_f32(dev_ptr(dy, name="dy"), dev_ptr(y_gate, name="y_gate", allow_none=True),
                                            gate_act, dev_ptr(weight, name="weight"), dev_ptr(dx, name="dx"),
                                            dev_ptr(x_gate, name="x_gate", allow_none=True), x_gate_act,
                                            dev_ptr(ws, name="workspace"), B, Cin, H, W, kh, kw, sh, sw, Cout,
                                            int(prep is not None), stream_ptr(dy.device)), "trl_conv_bwd_input_nhwc_f32")
    return dx


def transpose_bpc(x, B, P, Cc, y_gate=None, gate_act=ACT_NONE, gate_like_in=False):
    out = torch.empty((B, Cc, P), dtype=torch.float32, device=x.device)
    if y_gate is not None:                                 # out *= act'(y_gate), y_gate laid out like out (or like x)
        if y_gate.numel() != out.numel():
            raise TrlError("transpose_bpc: gate does not match the output")
        check(lib().trl_transpose_bpc_gate_f32(dev_ptr(x, name="x"), dev_ptr(y_gate, name="y_gate"), gate_act,
                                               int(bool(gate_like_in)), dev_ptr(out, name="out"), B, P, Cc,
                                               stream_ptr(x.device)),
              "trl_transpose_bpc_gate_f32")
        return out
    check(lib().trl_transpose_bpc_f32(dev_ptr(x, name="x"), dev_ptr(out, name="out"), B, P, Cc,
                                      stream_ptr(x.device)), "trl_transpose_bpc_f32")
    return out


def _ring_args(ring):
    """(ring (slots, 3) float64, device update counter) -> the three C arguments; None -> no filing."""
    if ring is None:
        return None, 0, None
    rows, counter = ring
    if rows.dim() != 2 or int(rows.shape[1]) != 3 or not rows.is_contiguous():
        raise TrlError("loss ring: (slots, 3) float64 rows")
    return dev_ptr(rows, torch.float64, "ring"), int(rows.shape[0]), dev_ptr(counter, torch.float64, "counter")


def _acts_ptrs(acts):
    """(int64 pointer, float32 pointer) of the stored actions -- exactly one is non-null."""
    if acts.dtype == torch.float32:
        return None, dev_ptr(acts, name="acts")
    return dev_ptr(acts, torch.int64, "acts"), None


def dqn_td_loss(q, acts, q_next, rew, term, gamma, sums, ring=None):
    """`acts` int64, or float32 as the replay buffer stores them; `ring` = (rows (slots, 3) float64, device update counter):
    the three sums are also filed into row (counter mod slots)."""
    B, A = int(q.shape[0]), int(q.shape[1])
    dq = torch.empty_like(q)
    ai, af = _acts_ptrs(acts)
    rp, slots, cp = _ring_args(ring)
    check(lib().trl_dqn_td_loss_f32(dev_ptr(q, name="q"), ai, af, dev_ptr(q_next, name="q_next"), dev_ptr(rew, name="rew"),
                                    dev_ptr(term, name="term"), float(gamma), B, A, dev_ptr(dq, name="dq"),
                                    dev_ptr(sums, torch.float64, "sums"), rp, slots, cp, stream_ptr(q.device)),
          "trl_dqn_td_loss_f32")
    return dq


def dqn_head_supported(H, A):
    return bool(lib().trl_dqn_head_supported(int(H), int(A)))


def dqn_head_workspace(H, A, device):
    """Zeroed workspace of `dqn_head` (keep it: it carries the kernel's arrival counter from launch to launch)."""
    return torch.zeros(int(lib().trl_dqn_head_workspace(int(H), int(A))), dtype=torch.uint8, device=device)


def dqn_head(h, h_next, w, bias, w_t, bias_t, acts, rew, term, gamma, dw, db, sums, workspace, ring=None, want_q=False):
    """The DQN head's forward, loss and backward in one launch (include/trl_hip.h K14b).  Fills dw (A, H), db (A), sums;
    returns dh (B, H), or (dh, q, q_next) with `want_q`."""
    B, H, A = int(h.shape[0]), int(h.shape[1]), int(w.shape[0])
    dh = torch.empty_like(h)
    q = torch.empty((B, A), dtype=torch.float32, device=h.device) if want_q else None
    qn = torch.empty((B, A), dtype=torch.float32, device=h.device) if want_q else None
    ai, af = _acts_ptrs(acts)
    rp, slots, cp = _ring_args(ring)
    check(lib().trl_dqn_head_f32(dev_ptr(h, name="h"), dev_ptr(h_next, name="h_next"), dev_ptr(w, name="w"),
                                 dev_ptr(bias, name="bias", allow_none=True), dev_ptr(w_t, name="w_t"),
                                 dev_ptr(bias_t, name="bias_t", allow_none=True), ai, af, dev_ptr(rew, name="rew"),
                                 dev_ptr(term, name="term"), float(gamma), B, H, A, dev_ptr(dh, name="dh"),
                                 dev_ptr(dw, name="dw"), dev_ptr(db, name="db"), dev_ptr(q, name="q", allow_none=True),
                                 dev_ptr(qn, name="q_next", allow_none=True), dev_ptr(sums, torch.float64, "sums"),
                                 rp, slots, cp, dev_ptr(workspace, torch.uint8, "workspace"), stream_ptr(h.device)),
          "trl_dqn_head_f32")
    return (dh, q, qn) if want_q else dh


def quantile_huber(q, acts, q_next, rew, term, gamma, A, Q, sums, ring=None):
    B = int(q.shape[0])
    dq = torch.empty_like(q)
    ws = torch.empty(2 * B, dtype=torch.float64, device=q.device)
    ai, af = _acts_ptrs(acts)
    rp, slots, cp = _ring_args(ring)
    check(lib().trl_quantile_huber_f32(dev_ptr(q, name="q"), ai, af, dev_ptr(q_next, name="q_next"), dev_ptr(rew, name="rew"),
                                       dev_ptr(term, name="term"), float(gamma), B, A, Q, dev_ptr(dq, name="dq"),
                                       dev_ptr(ws, torch.float64, "ws"), dev_ptr(sums, torch.float64, "sums"),
                                       rp, slots, cp, stream_ptr(q.device)), "trl_quantile_huber_f32")
    return dq


def eps_greedy(q, A, Q, u, rand_act, epsilon, ring_row=None, n_rows=0):
    N = int(q.shape[0])
    action = torch.empty(N, dtype=torch.int64, device=q.device)
    check(lib().trl_eps_greedy_i64(dev_ptr(q, name="q"), N, A, Q, dev_ptr(u, name="u", allow_none=True),
                                   dev_ptr(rand_act, torch.int64, "rand_act", allow_none=True), float(epsilon),
                                   dev_ptr(action, torch.int64, "action"),
                                   dev_ptr(ring_row, torch.int64, "ring_row", allow_none=True), int(n_rows),
                                   stream_ptr(q.device)), "trl_eps_greedy_i64")
    return action


def dqn_act_ok(h, w):
    return (h.dim() == 2 and w.dim() == 2 and h.is_contiguous() and w.is_contiguous() and h.data_ptr() % 16 == 0
            and w.data_ptr() % 16 == 0 and bool(lib().trl_dqn_act_supported(int(h.shape[1]), int(w.shape[0]))))


def dqn_act(h, w, bias, u, rand_act, epsilon, want_q=True, ring_row=None, n_rows=0):
    """(q (N, A) or None, action (N,) int64): the A <= 8 wide head on the last hidden activations + the epsilon-greedy
    action, one launch (include/trl_hip.h trl_dqn_act_f32); u / rand_act None: greedy."""
    N, H, A = int(h.shape[0]), int(h.shape[1]), int(w.shape[0])
    q = torch.empty((N, A), dtype=torch.float32, device=h.device) if want_q else None
    action = torch.empty(N, dtype=torch.int64, device=h.device)
    check(lib().trl_dqn_act_f32(dev_ptr(h, name="h"), dev_ptr(w, name="w"), dev_ptr(bias, name="bias", allow_none=True), N, H, A,
                                dev_ptr(u, name="u", allow_none=True), dev_ptr(rand_act, torch.int64, "rand_act", allow_none=True),
                                float(epsilon), dev_ptr(q, name="q", allow_none=True), dev_ptr(action, torch.int64, "action"),
                                dev_ptr(ring_row, torch.int64, "ring_row", allow_none=True), int(n_rows),
                                stream_ptr(h.device)), "trl_dqn_act_f32")
    return q, action


def synth_frames_step(frames, acts, t_env, seed_base, horizon, A, next_obs, rewards, dones):
    N, Cc, HW = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2]) * int(frames.shape[3])
    check(lib().trl_synth_frames_step_u8(dev_ptr(frames, torch.uint8, "frames"), dev_ptr(acts, torch.int64, "acts"),
                                         dev_ptr(t_env, torch.int32, "t_env"), int(seed_base), int(horizon), int(A),
                                         dev_ptr(next_obs, torch.uint8, "next_obs", allow_none=True),
                                         dev_ptr(rewards, name="rewards"), dev_ptr(dones, name="dones"), N, Cc, HW,
                                         stream_ptr(frames.device)), "trl_synth_frames_step_u8")


def synth_frames_reset(frames, t_env, seed_base, mask, ring_row=None, n_rows=0):
    N, Cc, HW = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2]) * int(frames.shape[3])
    check(lib().trl_synth_frames_reset_u8(dev_ptr(frames, torch.uint8, "frames"), dev_ptr(t_env, torch.int32, "t_env"),
                                          int(seed_base), dev_ptr(mask, torch.uint8, "mask", allow_none=True),
                                          dev_ptr(ring_row, torch.int64, "ring_row", allow_none=True), int(n_rows), N, Cc, HW,
                                          stream_ptr(frames.device)), "trl_synth_frames_reset_u8")


def synth_frames_collect(frames, acts, t_env, seed_base, horizon, A, ring, ring_row, step_rewards, step_dones, book=None):
    """ring = (obs, next_obs, acts, rewards, terminals, time_limits) ring TENSORS (rows, N, ...): the step files its
    transition into row ring_row[0] (device int64).  book = (cur_step, ep_return, max_frames, mask, epoch_reward, ep_count,
    ep_log, step): collector_bookkeep's arguments -- bookkeeping and the reset of the envs that ended in the same launch."""
    N, Cc, HW = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2]) * int(frames.shape[3])
    r_obs, r_next, r_acts, r_rew, r_done, r_tl = ring
    rows = int(r_obs.shape[0])
    for t in ring:
        if int(t.shape[0]) != rows or int(t.shape[1]) != N or not t.is_contiguous():
            raise TrlError("synth_frames_collect: ring tensors must be contiguous (rows, N, ...)")
    if book is not None:
        cur_step, ep_return, max_frames, mask, epoch_reward, ep_count, ep_log, step = book
        bk = [dev_ptr(cur_step, torch.int32, "cur_step"), dev_ptr(ep_return, name="ep_return"), int(max_frames),
              dev_ptr(mask, torch.uint8, "mask"), dev_ptr(epoch_reward, torch.float64, "epoch_reward", allow_none=True),
              dev_ptr(ep_count, torch.int32, "ep_count"), dev_ptr(ep_log, name="ep_log"), int(ep_log.shape[0]), int(step)]
    else:
        bk = [None, None, 0, None, None, None, None, 0, 0]
    check(lib().trl_synth_frames_collect_u8(
        dev_ptr(frames, torch.uint8, "frames"), dev_ptr(acts, torch.int64, "acts"), dev_ptr(t_env, torch.int32, "t_env"),
        int(seed_base), int(horizon), int(A), dev_ptr(r_obs, torch.uint8, "ring obs"), dev_ptr(r_next, torch.uint8, "ring next_obs"),
        dev_ptr(r_acts, name="ring acts"), dev_ptr(r_rew, name="ring rewards"), dev_ptr(r_done, name="ring terminals"),
        dev_ptr(r_tl, name="ring time_limits"), dev_ptr(ring_row, torch.int64, "ring_row"), rows,
        dev_ptr(step_rewards, name="step_rewards"), dev_ptr(step_dones, name="step_dones"), *bk, N, Cc, HW,
        stream_ptr(frames.device)), "trl_synth_frames_collect_u8")


def gauss_explore(mean, logstd, eps, tanh_action, act=None, logp=None):
    N, A = int(mean.shape[0]), int(mean.shape[1])
    if act is None:
        act = torch.empty((N, A), dtype=torch.float32, device=mean.device)
    if logp is None:
        logp = torch.empty((N,), dtype=torch.float32, device=mean.device)
    check(lib().trl_gauss_explore_f32(dev_ptr(mean, name="mean"), dev_ptr(logstd, name="logstd"),
                                      dev_ptr(eps, name="eps", allow_none=True), dev_ptr(act, name="act"),
                                      dev_ptr(logp, name="logp"), N, A, int(bool(tanh_action)), stream_ptr(mean.device)),
          "trl_gauss_explore_f32")
    return act, logp


def onpolicy_bookkeep(rewards, dones, v_next, discount, terminals, cur_step, ep_return, max_frames, mask, any_flag,
                      epoch_reward, ep_count, ep_log, step):
    N = int(rewards.numel())
    check(lib().trl_onpolicy_bookkeep_f32(dev_ptr(rewards, name="rewards"), dev_ptr(dones, name="dones"),
                                          dev_ptr(v_next, name="v_next"), float(discount), dev_ptr(terminals, name="terminals"),
                                          dev_ptr(cur_step, torch.int32, "cur_step"), dev_ptr(ep_return, name="ep_return"),
                                          int(max_frames), dev_ptr(mask, torch.uint8, "mask"),
                                          dev_ptr(any_flag, torch.int32, "any_flag"),
                                          dev_ptr(epoch_reward, torch.float64, "epoch_reward"),
                                          dev_ptr(ep_count, torch.int32, "ep_count"), dev_ptr(ep_log, name="ep_log"),
                                          int(ep_log.shape[0]), int(step), N, stream_ptr(rewards.device)),
          "trl_onpolicy_bookkeep_f32")


def select_on_flag(flag, a, b, out):
    check(lib().trl_select_on_flag_f32(dev_ptr(flag, torch.int32, "flag"), dev_ptr(a, name="a"), dev_ptr(b, name="b"),
                                       dev_ptr(out, name="out"), int(out.numel()), stream_ptr(out.device)),
          "trl_select_on_flag_f32")
    return out


def select_on_mask(mask, a, b, out):
    """out = a when any byte of `mask` is set, else b (no host round trip)."""
    check(lib().trl_select_on_mask_f32(dev_ptr(mask, torch.uint8, "mask"), int(mask.numel()), dev_ptr(a, name="a"),
                                       dev_ptr(b, name="b"), dev_ptr(out, name="out"), int(out.numel()),
                                       stream_ptr(out.device)), "trl_select_on_mask_f32")
    return out


def norm_update_filt(x, state, out, clip, update):
    """One vector step of the running observation normaliser: state (2D+1) fp64 = mean | var | count."""
    N, D = int(x.shape[0]), int(x.shape[1])
    check(lib().trl_norm_update_filt_f32(dev_ptr(x, name="x"), dev_ptr(state, torch.float64, "state"),
                                         dev_ptr(out, name="out", allow_none=True), N, D, float(clip), int(bool(update)),
                                         stream_ptr(x.device)), "trl_norm_update_filt_f32")
    return out


def norm_batch_moments(x, sums):
    N, D = int(x.shape[0]), int(x.shape[1])
    check(lib().trl_norm_batch_moments_f64(dev_ptr(x, name="x"), N, D, dev_ptr(sums, torch.float64, "sums"),
                                           stream_ptr(x.device)), "trl_norm_batch_moments_f64")
    return sums


def norm_merge(state, sums, D):
    check(lib().trl_norm_merge_f64(dev_ptr(state, torch.float64, "state"), dev_ptr(sums, torch.float64, "sums"), int(D),
                                   stream_ptr(state.device)), "trl_norm_merge_f64")


def norm_filt(x, state, out, clip):
    N, D = int(x.shape[0]), int(x.shape[1])
    check(lib().trl_norm_filt_f32(dev_ptr(x, name="x"), dev_ptr(state, torch.float64, "state"), dev_ptr(out, name="out"),
                                  N, D, float(clip), stream_ptr(x.device)), "trl_norm_filt_f32")
    return out


def detac_losses(q1, q2, tq1, tq2, rew, term, qn, gamma, sums):
    """DDPG / TD3 losses; q2 / tq2 / qn may be None.  Returns (dq1, dq2, dqn)."""
    B = int(q1.numel())
    mk = lambda ref: torch.empty_like(ref) if ref is not None else None
    dq1, dq2, dqn = mk(q1), mk(q2), mk(qn)
    p = lambda t, n: dev_ptr(t, name=n, allow_none=True)
    check(lib().trl_detac_losses_f32(p(q1, "q1"), p(q2, "q2"), p(tq1, "tq1"), p(tq2, "tq2"), p(rew, "rew"), p(term, "term"),
                                     p(qn, "qn"), float(gamma), B, p(dq1, "dq1"), p(dq2, "dq2"), p(dqn, "dqn"),
                                     dev_ptr(sums, torch.float64, "sums"), stream_ptr(q1.device)), "trl_detac_losses_f32")
    return dq1, dq2, dqn


def noisy_action(act, eps, sigma, noise_clip=float("inf"), lo=-float("inf"), hi=float("inf"), out=None):
    if out is None:
        out = torch.empty_like(act)
    check(lib().trl_noisy_action_f32(dev_ptr(act, name="act"), dev_ptr(eps, name="eps"), float(sigma), float(noise_clip),
                                     float(lo), float(hi), dev_ptr(out, name="out"), int(act.numel()),
                                     stream_ptr(act.device)), "trl_noisy_action_f32")
    return out


def frame_stream_append(stacks, stream, head, mask, n_frames):
    """stacks (N, C, H, W) uint8; stream (S, N, H*W) uint8; head (N) int32."""
    N, Cc = int(stacks.shape[0]), int(stacks.shape[1])
    HW = int(stacks[0, 0].numel())
    check(lib().trl_frame_stream_append_u8(dev_ptr(stacks, torch.uint8, "stacks"), dev_ptr(stream, torch.uint8, "stream"),
                                           dev_ptr(head, torch.int32, "head"),
                                           dev_ptr(mask, torch.uint8, "mask", allow_none=True), int(n_frames),
                                           int(stream.shape[0]), N, Cc, HW, stream_ptr(stacks.device)),
          "trl_frame_stream_append_u8")


def frame_stream_gather(stream, pos, row_idx, shift, frame_shape, head, overrun, out=None):
    """-> (n_rows * N, C, H, W) uint8 stacks rebuilt from the per-env frame stream (into `out` when given)."""
    S, N = int(stream.shape[0]), int(stream.shape[1])
    Cc, H, W = frame_shape
    nr = int(row_idx.numel())
    if out is None:
        out = torch.empty((nr * N, Cc, H, W), dtype=torch.uint8, device=stream.device)
    elif out.dtype != torch.uint8 or out.numel() != nr * N * Cc * H * W or not out.is_contiguous():
        raise TrlError("frame_stream_gather: out must be a contiguous uint8 tensor of %d stacks" % (nr * N))
    check(lib().trl_frame_stream_gather_u8(dev_ptr(stream, torch.uint8, "stream"), dev_ptr(pos, torch.int32, "pos"),
                                           dev_ptr(row_idx, torch.int64, "row_idx"), nr, int(shift),
                                           dev_ptr(out, torch.uint8, "out"), dev_ptr(head, torch.int32, "head"),
                                           dev_ptr(overrun, torch.int32, "overrun"), S, N, Cc, H * W,
                                           stream_ptr(stream.device)), "trl_frame_stream_gather_u8")
    return out
