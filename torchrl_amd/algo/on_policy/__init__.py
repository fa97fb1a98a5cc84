from .on_rl_algo import OnRLAlgo
from .a2c import A2C
from .ppo import PPO
from .v_mpo import VMPO
from .trpo import TRPO
from .not_built import Reinforce
