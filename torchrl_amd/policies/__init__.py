from .continuous_policy import (GuassianContPolicy, GuassianContPolicyBasicBias, GuassianContPolicyBase,
                                FixGuassianContPolicy, DetContPolicy, UniformPolicyContinuous)
from .discrete_policies import EpsilonGreedyDQNDiscretePolicy, EpsilonGreedyQRDQNDiscretePolicy, CategoricalDisPolicy
from .distribution import TanhNormal
