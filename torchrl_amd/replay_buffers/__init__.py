from .base import BaseReplayBuffer
from .on_policy import OnPolicyReplayBuffer
