"""Env factories with the reference's signatures (torchrl/env/get_env.py:39-87).

The synthetic on-GPU ids (no gym / MuJoCo / ALE in the image, and real physics is not the benchmark) plus two
pure-Python host tasks (`PyPendulum-v0`, `PyCartPole-v0`) behind `VecEnv` / `SubProcVecEnv`."""
from .base_wrapper import NormObs
from .py_envs import CartPoleEnv, PendulumEnv
from .subproc_vecenv import SubProcVecEnv
from .synth import SynthVecEnv, SynthFrameVecEnv, SYNTH_IDS, SYNTH_FRAME_IDS
from .vecenv import VecEnv

# pure-Python tasks stepped on the host (the reference's gym ids are not installable here)
HOST_IDS = {"PyPendulum-v0": PendulumEnv, "PyCartPole-v0": CartPoleEnv}


def _check(env_id, env_param):
    if env_id not in SYNTH_IDS and env_id not in SYNTH_FRAME_IDS and env_id not in HOST_IDS:
        raise ValueError("unknown env id %r: torchrl_amd ships the synthetic ids %s and the host ids %s"
                         % (env_id, sorted(SYNTH_IDS) + sorted(SYNTH_FRAME_IDS), sorted(HOST_IDS)))
    if "rew_norm" in env_param:
        raise NotImplementedError("rew_norm (NormRet wrapper) is not built")


def get_vec_env(env_id, env_param, vec_env_nums, device=None, index_offset=0, total_env_nums=None):
    _check(env_id, env_param)
    if env_id in HOST_IDS:                                   # host Python envs: the collectors bridge them to the device
        if env_param.get("obs_norm", False):
            raise NotImplementedError("obs_norm on host ids: wrap NormObs(HostEnvBridge(env, device)) explicitly")
        return VecEnv(vec_env_nums, HOST_IDS[env_id], ())
    if env_id in SYNTH_FRAME_IDS:
        return SynthFrameVecEnv(vec_env_nums, reward_scale=env_param.get("reward_scale", 1), device=device,
                                index_offset=index_offset, total_env_nums=total_env_nums, **SYNTH_FRAME_IDS[env_id])
    env = SynthVecEnv(vec_env_nums, reward_scale=env_param.get("reward_scale", 1), device=device,
                      index_offset=index_offset, total_env_nums=total_env_nums, **SYNTH_IDS[env_id])
    if env_param.get("obs_norm", False):                     # env/get_env.py:75-76
        env = NormObs(env)
    return env


def get_subprocvec_env(env_id, env_param, vec_env_nums, proc_nums, **kwargs):
    """Host ids: the envs are stepped in `proc_nums` spawned worker processes (env/subproc_vecenv.py); the synthetic
    ids live on the GPU, where process-parallel stepping is what the kernels replace."""
    _check(env_id, env_param)
    if env_id in HOST_IDS:
        return SubProcVecEnv(proc_nums, vec_env_nums, HOST_IDS[env_id], ())
    return get_vec_env(env_id, env_param, vec_env_nums, **kwargs)


def get_env(env_id, env_param, **kwargs):
    return get_vec_env(env_id, env_param, 1, **kwargs)
