from .off_rl_algo import OffRLAlgo
from .twin_sac_q import TwinSACQ
from .dqn import DQN, QRDQN
from .det_ac import DDPG, TD3
