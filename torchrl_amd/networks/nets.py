"""Net / QNet heads (torchrl/networks/nets.py:13-68) + the flat-parameter view
the HIP kernels consume.

A `Net` whose trunk is an `MLPBase` with two equal hidden widths, Tanh or ReLU,
no LayerNorm and no extra hidden layers is an "MLP2" block (include/trl_hip.h):
`mlp2_spec()` describes it and `flatten_mlp2()` re-homes its parameters as
views into ONE contiguous fp32 buffer  W1 b1 W2 b2 W3 b3 [logstd]  so kernels
read/write the very storage `state_dict()` / `parameters()` expose.

`forward` on a GPU MLP2 under no_grad runs trl_mlp2_forward_f32.  With autograd
enabled (algorithms outside this repo's hot-path scope) it is the plain
nn.Module graph.
"""
import torch
import torch.nn as nn

from . import init
from .base import MLPBase
from .. import _C

_ACT_CODE = {nn.Tanh: _C.ACT_TANH, nn.ReLU: _C.ACT_RELU}


# Parameters that a launch sequence on ANOTHER stream is still stepping (the critic's half of a PPO epoch runs beside the
# next rollout, algo/on_policy/ppo.py): net -> the event that sequence ends with.  Whoever is about to read such a net's
# parameters on the current stream waits for the event first (`settle`); weak keys: nothing is kept alive, nothing is
# pickled or deep-copied with the module.
import weakref
_PENDING = weakref.WeakKeyDictionary()


def set_pending(net, event):
    if event is None:
        _PENDING.pop(net, None)
    else:
        _PENDING[net] = event


def pending_event(net):
    return _PENDING.get(net)


def settle(net, device=None):
    """Make the current stream wait for whatever is still writing `net`'s parameters on another stream."""
    ev = _PENDING.get(net)
    if ev is not None:
        torch.cuda.current_stream(device).wait_event(ev)


class Net(nn.Module):
    def __init__(self, output_shape, base_type, append_hidden_shapes=[],
                 append_hidden_init_func=init.basic_init, net_last_init_func=init.uniform_init,
                 activation_func=nn.ReLU, add_ln=False, **kwargs):
        super().__init__()
        self.base = base_type(activation_func=activation_func, add_ln=add_ln, **kwargs)
        self.add_ln = add_ln
        self.activation_func = activation_func
        width = self.base.output_shape
        layers = []
        for nxt in append_hidden_shapes:
            fc = nn.Linear(width, nxt)
            append_hidden_init_func(fc)
            layers += [fc, activation_func()]
            if add_ln:
                layers.append(nn.LayerNorm(nxt))
            width = nxt
        last = nn.Linear(width, output_shape)
        net_last_init_func(last)
        layers.append(last)
        self.append_fcs = layers
        self.seq_append_fcs = nn.Sequential(*layers)
        self.out_dim = output_shape
        self._flat = None

    # ---- MLP2 description / flat storage ----
    def mlp2_spec(self):
        """(D, H, O, act_code) if this net is an MLP2 block the kernels support, else None."""
        b = self.base
        if not isinstance(b, MLPBase) or self.add_ln or len(self.append_fcs) != 1:
            return None
        if len(b.hidden_shapes) != 2 or b.hidden_shapes[0] != b.hidden_shapes[1]:
            return None
        if b.activation_func not in _ACT_CODE or b.last_activation_func is not b.activation_func:
            return None
        return (b.input_dim, b.hidden_shapes[0], self.out_dim, _ACT_CODE[b.activation_func])

    def _mlp2_param_list(self):
        lin = [m for m in self.base.seq_fcs if isinstance(m, nn.Linear)] + [self.append_fcs[-1]]
        out = []
        for l in lin:
            out += [l.weight, l.bias]
        return out

    def flat_params(self):
        """Contiguous fp32 buffer aliasing this net's parameters (built on first use and
        re-built if `.to()` / `load_state_dict` moved the storage)."""
        plist = self._mlp2_param_list() + self._extra_flat_params()
        flat = self._flat
        ok = flat is not None and flat.device == plist[0].device
        if ok:
            off = 0
            for p in plist:
                if p.data_ptr() != flat.data_ptr() + 4 * off or not p.is_contiguous():
                    ok = False
                    break
                off += p.numel()
        if not ok:
            flat = flatten_into(plist)
            self._flat = flat
        return flat

    def _extra_flat_params(self):
        return []

    # (whoever saves or loads the parameters reads / writes them on the current stream: settle first, like `forward`)
    def _settle_params(self):
        if _PENDING and self in _PENDING:
            p = next(self.parameters(), None)
            settle(self, p.device if p is not None and p.is_cuda else None)

    def state_dict(self, *args, **kwargs):
        self._settle_params()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._settle_params()
        return super().load_state_dict(*args, **kwargs)

    def forward(self, x):
        spec = self.mlp2_spec()
        needs_graph = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if x.is_cuda and _PENDING:
            settle(self, x.device)
        if x.is_cuda and not needs_graph:
            lead = x.shape[:-1]
            if spec is not None and _C.lib().trl_mlp2_forward_supported(spec[0], spec[1], spec[2]):
                D, H, O, act = spec                                 # fused 2-layer kernel
                x2 = x.reshape(-1, D).float().contiguous()
                return _C.mlp2_forward(self.flat_params(), x2, D, H, O, act).reshape(tuple(lead) + (O,))
            from .. import ops
            try:
                layers, act = ops.linear_layers(self), ops.act_code(self)
            except _C.TrlError:
                layers = None                                       # LayerNorm / conv trunk / other activations: torch modules
            if layers is not None and isinstance(self.base, MLPBase):
                x2 = x.reshape(-1, int(layers[0][0].shape[1])).float().contiguous()
                out, _ = ops.mlp_forward([(w.detach(), b.detach()) for w, b in layers], x2, act)   # dense-layer kernels
                return out.reshape(tuple(lead) + (self.out_dim,))
        if x.is_cuda:
            _C.note_eager(type(self).__name__ + ".forward", "autograd is on" if needs_graph else "no kernel for this trunk")
        return self.seq_append_fcs(self.base(x))


def flatten_into(plist):
    """Copy parameters into one buffer and turn them into views of it."""
    total = sum(p.numel() for p in plist)
    flat = torch.empty(total, dtype=torch.float32, device=plist[0].device)
    off = 0
    with torch.no_grad():
        for p in plist:
            n = p.numel()
            flat[off:off + n].copy_(p.detach().reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
            off += n
    return flat


class FlattenNet(Net):
    def forward(self, input):
        return super().forward(torch.cat(input, dim=-1))


class QNet(Net):
    def forward(self, input):
        assert len(input) == 2, "Q Net only get observation and action"
        state, action = input
        return super().forward(torch.cat([state, action], dim=-1))


class ZeroNet(nn.Module):
    def forward(self, x):
        return torch.zeros(1)
