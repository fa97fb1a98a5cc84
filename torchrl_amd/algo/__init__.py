from .rl_algo import RLAlgo
from .on_policy import OnRLAlgo, A2C, PPO
from . import utils
