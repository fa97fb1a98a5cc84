from .continuous_policy import (GuassianContPolicy, GuassianContPolicyBasicBias, GuassianContPolicyBase,
                                FixGuassianContPolicy, UniformPolicyContinuous)
from .distribution import TanhNormal
