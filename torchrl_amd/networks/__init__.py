from .base import MLPBase, CNNBase, calc_next_shape
from .nets import Net, QNet, FlattenNet, ZeroNet, flatten_into
from . import init
from .init import layer_init, basic_init, uniform_init, orthogonal_init   # reference: `from .init import *` (networks/__init__.py:3)
