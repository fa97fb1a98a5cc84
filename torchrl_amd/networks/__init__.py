from .base import MLPBase, CNNBase, calc_next_shape
from .nets import Net, QNet, FlattenNet, ZeroNet, flatten_into
from . import init
