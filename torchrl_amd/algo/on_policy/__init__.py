from .on_rl_algo import OnRLAlgo
from .a2c import A2C
from .ppo import PPO
from .v_mpo import VMPO
from .not_built import TRPO, Reinforce
