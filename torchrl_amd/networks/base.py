"""MLP / CNN trunks with the reference's constructor signatures and state_dict
names (torchrl/networks/base.py:8-107): Linear -> activation after EVERY hidden
layer (the last hidden activation is `last_activation_func`, default the same
class), optional LayerNorm.  The modules only define parameters and structure;
the arithmetic on the hot path runs in the HIP kernels (see nets.Net.forward)."""
import numpy as np
import torch.nn as nn

from . import init


class MLPBase(nn.Module):
    def __init__(self, input_shape, hidden_shapes, activation_func=nn.ReLU,
                 init_func=init.basic_init, add_ln=False, last_activation_func=None):
        super().__init__()
        self.activation_func = activation_func
        self.add_ln = add_ln
        self.last_activation_func = last_activation_func if last_activation_func is not None else activation_func
        width = int(np.prod(input_shape))
        self.input_dim = width
        self.hidden_shapes = list(hidden_shapes)
        self.output_shape = width
        layers = []
        for nxt in hidden_shapes:
            fc = nn.Linear(width, nxt)
            init_func(fc)
            layers += [fc, activation_func()]
            if add_ln:
                layers.append(nn.LayerNorm(nxt))
            width = nxt
            self.output_shape = nxt
        layers.pop(-1)                                   # same quirk as the reference (base.py:39-40)
        layers.append(self.last_activation_func())
        self.fcs = layers
        self.seq_fcs = nn.Sequential(*layers)

    def forward(self, x):
        return self.seq_fcs(x)


def calc_next_shape(input_shape, conv_info):
    out_channels, kernel_size, stride, padding = conv_info
    _, h, w = input_shape
    h = int((h + 2 * padding[0] - (kernel_size[0] - 1) - 1) / stride[0] + 1)
    w = int((w + 2 * padding[1] - (kernel_size[1] - 1) - 1) / stride[1] + 1)
    return (out_channels, h, w)


class CNNBase(nn.Module):
    """Conv trunk (torchrl/networks/base.py:59-107).  Structure only in this round:
    the Atari-shaped configs (SURVEY.md section 8 cfg 5) are a later row."""

    def __init__(self, input_shape, hidden_shapes, activation_func=nn.ReLU,
                 init_func=init.basic_init, add_ln=False, last_activation_func=None):
        super().__init__()
        shape = input_shape
        channels = input_shape[0]
        self.add_ln = add_ln
        self.activation_func = activation_func
        self.last_activation_func = last_activation_func if last_activation_func is not None else activation_func
        self.output_shape = shape[0] * shape[1] * shape[2]
        layers = []
        for info in hidden_shapes:
            out_channels, kernel_size, stride, padding = info
            conv = nn.Conv2d(channels, out_channels, kernel_size, stride, padding)
            init_func(conv)
            layers += [conv, activation_func()]
            channels = out_channels
            shape = calc_next_shape(shape, info)
            if add_ln:
                layers.append(nn.LayerNorm(shape[1:]))
            self.output_shape = shape[0] * shape[1] * shape[2]
        layers.pop(-1)
        layers.append(self.last_activation_func())
        self.convs = layers
        self.seq_convs = nn.Sequential(*layers)

    def forward(self, x):
        lead = x.size()[:-3]
        x = x.reshape((int(np.prod(lead)) if len(lead) else 1,) + tuple(x.size()[-3:]))
        return self.seq_convs(x).view(tuple(lead) + (-1,))
