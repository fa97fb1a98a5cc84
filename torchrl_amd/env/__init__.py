from .get_env import get_env, get_vec_env, get_subprocvec_env
from .synth import SynthVecEnv, SynthFrameVecEnv

VecEnv = SynthVecEnv
SubProcVecEnv = SynthVecEnv
