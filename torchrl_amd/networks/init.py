"""Layer initialisers with the reference's exact conventions
(torchrl/networks/init.py:5-47): hidden layers draw U(+-sqrt(1/size[0])) where
size[0] is nn.Linear's out_features (the reference's "fan-in", its Q9 quirk),
biases are the constant 0.1; heads draw U(+-3e-3) for weight and bias."""
import numpy as np
import torch.nn as nn


def _fanin_init(tensor, alpha=0):
    dims = tensor.size()
    if len(dims) == 2:
        fan = dims[0]
    elif len(dims) > 2:
        fan = int(np.prod(dims[1:]))
    else:
        raise Exception("Shape must be have dimension at least 2.")
    limit = float(np.sqrt(1.0 / ((1 + alpha * alpha) * fan)))
    return tensor.data.uniform_(-limit, limit)


def _uniform_init(tensor, param=3e-3):
    return tensor.data.uniform_(-param, param)


def _constant_bias_init(tensor, constant=0.1):
    tensor.data.fill_(constant)


def layer_init(layer, weight_init=_fanin_init, bias_init=_constant_bias_init):
    weight_init(layer.weight)
    bias_init(layer.bias)


def basic_init(layer):
    layer_init(layer, _fanin_init, _constant_bias_init)


def uniform_init(layer):
    layer_init(layer, _uniform_init, _uniform_init)


def _orthogonal_init(tensor, gain=np.sqrt(2)):
    nn.init.orthogonal_(tensor, gain=gain)


def orthogonal_init(layer, scale=np.sqrt(2), constant=0):
    layer_init(layer, lambda w: _orthogonal_init(w, gain=scale), lambda b: _constant_bias_init(b, 0))
