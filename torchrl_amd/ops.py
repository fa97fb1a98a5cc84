"""Dense-network forward / backward on the generic fp32 MFMA layer kernels (k_gemm.hip).

An MLP here is what the reference builds from MLPBase + Net
(torchrl/networks/base.py:8-44, nets.py:13-52): hidden Linear layers each followed by the
activation, then a linear head.  `mlp_forward` keeps the layer outputs (the "tape");
`mlp_backward` walks it back: dW / db of every layer straight into views of a flat
gradient buffer, and optionally d(input).  act'(.) is applied inside the kernels through the
stored layer OUTPUTS, so no pre-activation tensors are kept.
"""
import torch
import torch.nn as nn

from . import _C

ACT_OF = {nn.Tanh: _C.ACT_TANH, nn.ReLU: _C.ACT_RELU}


def linear_layers(net):
    """[(weight, bias), ...] of a torchrl_amd.networks.Net whose trunk is an MLPBase without LayerNorm."""
    mods = [m for m in net.base.seq_fcs if isinstance(m, nn.Linear)] + \
           [m for m in net.seq_append_fcs if isinstance(m, nn.Linear)]
    return [(m.weight, m.bias) for m in mods]


def act_code(net):
    a = net.base.activation_func
    if a not in ACT_OF or net.base.add_ln or net.base.last_activation_func is not a:
        raise _C.TrlError("dense kernels support Tanh / ReLU MLPs without LayerNorm (got %s)" % a)
    return ACT_OF[a]


class Tape:
    __slots__ = ("x", "outs", "layers", "act")


def mlp_forward(layers, x, act):
    """layers: [(W, b), ...] (nn.Linear layout); returns (out, tape)."""
    t = Tape()
    t.x, t.layers, t.act, t.outs = x, layers, act, []
    h = x
    for k, (w, b) in enumerate(layers):
        last = k == len(layers) - 1
        h = _C.linear_fwd(h, w, b, _C.ACT_NONE if last else act)
        t.outs.append(h)
    return h, t


def mlp_backward(tape, d_out, grads=None, need_input=False, workspace=None):
    """d_out: gradient w.r.t. the network output.  grads: [(dW_view, db_view), ...] to fill (or None
    to skip weight gradients).  Returns d(input) if need_input."""
    d = d_out
    n = len(tape.layers)
    for k in range(n - 1, -1, -1):
        w, _b = tape.layers[k]
        gate = None if k == n - 1 else tape.outs[k]                 # hidden outputs gate through act'
        inp = tape.x if k == 0 else tape.outs[k - 1]
        if grads is not None:
            _C.linear_bwd_weight(d, gate, tape.act, inp, dw=grads[k][0], db=grads[k][1], workspace=workspace)
        if k > 0 or need_input:
            d = _C.linear_bwd_input(d, gate, tape.act, w)
    return d if need_input else None
