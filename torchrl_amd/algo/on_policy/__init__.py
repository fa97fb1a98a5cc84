from .on_rl_algo import OnRLAlgo
from .a2c import A2C
from .ppo import PPO
from .not_built import TRPO, VMPO, Reinforce
