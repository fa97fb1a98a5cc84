from .rl_algo import RLAlgo
from .on_policy import OnRLAlgo, A2C, PPO, TRPO, VMPO, Reinforce
from .off_policy import OffRLAlgo, TwinSACQ, DQN, QRDQN, DDPG, TD3
from . import utils
