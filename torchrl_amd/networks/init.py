"""Layer initialisers with the reference's conventions (torchrl/networks/init.py:5-47).

What has to be reproduced exactly, because seeded runs and the golden fixtures depend on it:
  * hidden layers (`basic_init`): weight ~ U(-b, b) with b = sqrt(1 / fan), where `fan` is the FIRST dimension of a 2-d
    weight -- nn.Linear's out_features, the reference's "fan-in" (its Q9 quirk) -- and the product of the trailing
    dimensions for conv weights; bias = 0.1;
  * heads (`uniform_init`): weight and bias ~ U(-3e-3, 3e-3);
  * `orthogonal_init`: orthogonal weight with the given gain, zero bias.
One `uniform_` call per tensor, weight before bias, so the torch generator is consumed in the reference's order."""
import math

import torch.nn as nn


def _fan(shape):
    if len(shape) < 2:
        raise Exception("Shape must be have dimension at least 2.")
    return shape[0] if len(shape) == 2 else math.prod(shape[1:])


def _fanin_init(tensor, alpha=0):
    limit = math.sqrt(1.0 / ((1.0 + alpha * alpha) * _fan(tuple(tensor.size()))))
    return tensor.data.uniform_(-limit, limit)


def _uniform_init(tensor, param=3e-3):
    return tensor.data.uniform_(-param, param)


def _constant_bias_init(tensor, constant=0.1):
    tensor.data.fill_(constant)


def _orthogonal_init(tensor, gain=math.sqrt(2)):
    nn.init.orthogonal_(tensor, gain=gain)


def layer_init(layer, weight_init=_fanin_init, bias_init=_constant_bias_init):
    for tensor, fn in ((layer.weight, weight_init), (layer.bias, bias_init)):
        fn(tensor)


def _recipe(weight_init, bias_init):
    def init(layer):
        layer_init(layer, weight_init, bias_init)
    return init


basic_init = _recipe(_fanin_init, _constant_bias_init)
uniform_init = _recipe(_uniform_init, _uniform_init)


def orthogonal_init(layer, scale=math.sqrt(2), constant=0):
    layer_init(layer, lambda w: _orthogonal_init(w, gain=scale), lambda b: _constant_bias_init(b, 0))
