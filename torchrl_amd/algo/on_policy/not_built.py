"""Names the reference's example scripts import (the old single-env scripts) whose algorithms have no
kernel path in this build.  Importing works; constructing one fails loudly instead of silently
running somewhere else -- there is no CPU / autograd fallback in this package."""
from ... import _C


class _NotBuilt:
    _what = ""

    def __init__(self, *args, **kwargs):
        raise _C.TrlError("%s is not built in torchrl_amd (reference: %s): the HIP path covers PPO, A2C, TRPO, VMPO, TwinSACQ, "
                          "DQN, QRDQN, DDPG and TD3" % (type(self).__name__, self._what))


class Reinforce(_NotBuilt):
    _what = "torchrl/algo/on_policy/reinforce.py"
