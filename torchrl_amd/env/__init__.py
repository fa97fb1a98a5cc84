from .get_env import get_env, get_vec_env, get_subprocvec_env
from .synth import SynthVecEnv, SynthFrameVecEnv
from .base_wrapper import Normalizer, NormObs

VecEnv = SynthVecEnv
SubProcVecEnv = SynthVecEnv
from .vecenv import VecEnv, HostEnvBridge
from .subproc_vecenv import SubProcVecEnv
