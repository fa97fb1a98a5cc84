"""Env factories with the reference's signatures (torchrl/env/get_env.py:39-87).

Only the synthetic on-GPU ids are available (no gym / MuJoCo / ALE in the image,
and real physics is not the benchmark).  `get_subprocvec_env` returns the same
device env: process-parallel stepping is what the GPU engine replaces."""
from .base_wrapper import NormObs
from .synth import SynthVecEnv, SynthFrameVecEnv, SYNTH_IDS, SYNTH_FRAME_IDS


def _check(env_id, env_param):
    if env_id not in SYNTH_IDS and env_id not in SYNTH_FRAME_IDS:
        raise ValueError("unknown env id %r: torchrl_amd ships the synthetic ids %s"
                         % (env_id, sorted(SYNTH_IDS) + sorted(SYNTH_FRAME_IDS)))
    if "rew_norm" in env_param:
        raise NotImplementedError("rew_norm (NormRet wrapper) is not built")


def get_vec_env(env_id, env_param, vec_env_nums, device=None, index_offset=0, total_env_nums=None):
    _check(env_id, env_param)
    if env_id in SYNTH_FRAME_IDS:
        return SynthFrameVecEnv(vec_env_nums, reward_scale=env_param.get("reward_scale", 1), device=device,
                                index_offset=index_offset, total_env_nums=total_env_nums, **SYNTH_FRAME_IDS[env_id])
    env = SynthVecEnv(vec_env_nums, reward_scale=env_param.get("reward_scale", 1), device=device,
                      index_offset=index_offset, total_env_nums=total_env_nums, **SYNTH_IDS[env_id])
    if env_param.get("obs_norm", False):                     # env/get_env.py:75-76
        env = NormObs(env)
    return env


def get_subprocvec_env(env_id, env_param, vec_env_nums, proc_nums, **kwargs):
    return get_vec_env(env_id, env_param, vec_env_nums, **kwargs)


def get_env(env_id, env_param, **kwargs):
    return get_vec_env(env_id, env_param, 1, **kwargs)
