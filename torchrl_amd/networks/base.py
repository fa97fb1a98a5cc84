"""MLP / CNN trunks: parameter containers with the reference's constructor signatures and state_dict
names (torchrl/networks/base.py:8-107).  A trunk is a flat nn.Sequential `[layer, act, (LayerNorm)] * n` whose LAST
element is dropped and replaced by `last_activation_func()` (default: the same activation class) -- the reference's
construction (base.py:39-40), so with `add_ln` the final LayerNorm is what gets replaced.  The modules only define
parameters and structure; the arithmetic of the hot path runs in the HIP kernels (see nets.Net.forward, ops.py)."""
import math

import torch.nn as nn

from . import init


def _trunk(blocks, activation_func, last_activation_func, add_ln):
    """blocks: [(parameter layer, LayerNorm shape)] -> the flat module list described above."""
    seq = []
    for layer, norm_shape in blocks:
        seq.append(layer)
        seq.append(activation_func())
        if add_ln:
            seq.append(nn.LayerNorm(norm_shape))
    seq[-1] = last_activation_func()
    return seq


class _Trunk(nn.Module):
    def _configure(self, activation_func, add_ln, last_activation_func):
        self.activation_func = activation_func
        self.add_ln = add_ln
        self.last_activation_func = activation_func if last_activation_func is None else last_activation_func


class MLPBase(_Trunk):
    def __init__(self, input_shape, hidden_shapes, activation_func=nn.ReLU,
                 init_func=init.basic_init, add_ln=False, last_activation_func=None):
        super().__init__()
        self._configure(activation_func, add_ln, last_activation_func)
        self.hidden_shapes = list(hidden_shapes)
        self.input_dim = int(math.prod(input_shape)) if hasattr(input_shape, "__len__") else int(input_shape)
        widths = [self.input_dim] + self.hidden_shapes
        blocks = []
        for fan_in, fan_out in zip(widths[:-1], widths[1:]):
            fc = nn.Linear(fan_in, fan_out)
            init_func(fc)
            blocks.append((fc, fan_out))
        self.output_shape = widths[-1]
        self.fcs = _trunk(blocks, activation_func, self.last_activation_func, add_ln)
        self.seq_fcs = nn.Sequential(*self.fcs)

    def forward(self, x):
        return self.seq_fcs(x)


def calc_next_shape(input_shape, conv_info):
    """(C, H, W) after one conv described as [out_channels, kernel, stride, padding] (base.py:47-56)."""
    out_channels, kernel, stride, padding = conv_info
    spatial = tuple(int((size + 2 * p - (k - 1) - 1) / s + 1)
                    for size, k, s, p in zip(input_shape[1:], kernel, stride, padding))
    return (out_channels,) + spatial


class CNNBase(_Trunk):
    """Conv trunk (torchrl/networks/base.py:59-107); forward flattens whatever leads the (C, H, W) axes into one batch
    axis and restores it on the way out."""

    def __init__(self, input_shape, hidden_shapes, activation_func=nn.ReLU,
                 init_func=init.basic_init, add_ln=False, last_activation_func=None):
        super().__init__()
        self._configure(activation_func, add_ln, last_activation_func)
        shape = tuple(input_shape)
        blocks = []
        for info in hidden_shapes:
            conv = nn.Conv2d(shape[0], info[0], info[1], info[2], info[3])
            init_func(conv)
            shape = calc_next_shape(shape, info)
            blocks.append((conv, shape[1:]))
        self.output_shape = shape[0] * shape[1] * shape[2]
        self.convs = _trunk(blocks, activation_func, self.last_activation_func, add_ln)
        self.seq_convs = nn.Sequential(*self.convs)

    def forward(self, x):
        lead = tuple(x.shape[:-3])
        flat = x.reshape((math.prod(lead) if lead else 1,) + tuple(x.shape[-3:]))
        return self.seq_convs(flat).view(lead + (-1,))
