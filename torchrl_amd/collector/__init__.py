from .base import BaseCollector, VecCollector
from .on_policy import OnPolicyCollectorBase, VecOnPolicyCollector
