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
    __slots__ = ("x", "outs", "layers", "act", "last_act")


def mlp_forward(layers, x, act, last_act=None, keep=True):
    """layers: [(W, b), ...] (nn.Linear layout); returns (out, tape).  `last_act` (an ACT_* code) is applied to
    the head output inside the last layer's epilogue (deterministic policies: tanh(mlp(x))).  keep=False: no backward pass
    will follow (acting): the fused three-layer launch then leaves its hidden activations on chip."""
    if len(layers) == 3 and all(b is not None for _, b in layers[:2]) and \
            _C.mlp3_forward_ok(layers[0][0].shape[1], layers[0][0].shape[0], layers[1][0].shape[0], layers[2][0].shape[0]):
        outs, tapes = mlp_forward_group([layers], [x], act, last_act=last_act, keep=[bool(keep)])      # one fused launch
        return outs[0], tapes[0]
    t = Tape()
    t.x, t.layers, t.act, t.outs = x, layers, act, []
    t.last_act = _C.ACT_NONE if last_act is None else last_act
    h = x
    for k, (w, b) in enumerate(layers):
        last = k == len(layers) - 1
        h = _C.linear_fwd(h, w, b, t.last_act if last else act)
        t.outs.append(h)
    return h, t


def mlp_backward(tape, d_out, grads=None, need_input=False, workspace=None, plan=None, head_dx=None):
    """d_out: gradient w.r.t. the network output.  grads: [(dW_view, db_view), ...] to fill (or None
    to skip weight gradients).  Returns d(input) if need_input.  plan (_C.FoldPlan): the weight gradients are only
    final after `plan.run()` (one fold launch for the whole pass).  head_dx (with plan): the gradient at the last hidden
    layer, ALREADY gated by that layer's act' (somebody computed it along the way): the last layer's input-gradient launch
    is skipped."""
    if plan is not None:
        out = mlp_backward_group([tape], [d_out], None if grads is None else [grads], need_input, plan=plan,
                                 head_dx=None if head_dx is None else [head_dx])
        return out[0] if need_input else None
    d = d_out
    n = len(tape.layers)
    for k in range(n - 1, -1, -1):
        w, _b = tape.layers[k]
        last = k == n - 1
        gate_act = tape.last_act if last else tape.act
        gate = None if (last and gate_act == _C.ACT_NONE) else tape.outs[k]   # outputs gate through act'
        inp = tape.x if k == 0 else tape.outs[k - 1]
        if grads is not None:
            _C.linear_bwd_weight(d, gate, gate_act, inp, dw=grads[k][0], db=grads[k][1], workspace=workspace)
        if k > 0 or need_input:
            d = _C.linear_bwd_input(d, gate, gate_act, w)
    return d if need_input else None


def mlp_forward_group(layers_list, xs, act, last_act=None, keep=None):
    """G same-shaped MLPs (or one MLP listed several times) on G inputs, one launch per layer instead of G:
    returns ([out_g], [tape_g]); every tape works with `mlp_backward` / `mlp_backward_group`.
    keep[g] = False: network g's hidden activations are not needed (no backward pass, e.g. target networks); networks
    of the shape D -> 256 -> 256 -> O then run through ONE fused launch that leaves their hidden layers on chip."""
    G = len(layers_list)
    code_last = _C.ACT_NONE if last_act is None else last_act
    ls0 = layers_list[0]
    if len(ls0) == 3 and _C.mlp3_forward_ok(ls0[0][0].shape[1], ls0[0][0].shape[0], ls0[1][0].shape[0], ls0[2][0].shape[0]) and \
            all(b is not None for ls in layers_list for _, b in ls[:2]):
        keep = [True] * G if keep is None else list(keep)
        res = _C.mlp3_forward_group(layers_list, xs, act, code_last, keep)
        tapes, outs = [], []
        for g, (h1, h2, y) in enumerate(res):
            t = Tape()
            t.x, t.layers, t.act, t.last_act = xs[g], layers_list[g], act, code_last
            t.outs = [h1, h2, y]
            tapes.append(t)
            outs.append(y)
        return outs, tapes
    tapes = []
    for g in range(G):
        t = Tape()
        t.x, t.layers, t.act, t.outs = xs[g], layers_list[g], act, []
        t.last_act = _C.ACT_NONE if last_act is None else last_act
        tapes.append(t)
    hs = list(xs)
    n = len(layers_list[0])
    for k in range(n):
        code = tapes[0].last_act if k == n - 1 else act
        hs = _C.linear_fwd_group(hs, [ls[k][0] for ls in layers_list], [ls[k][1] for ls in layers_list], code)
        for t, h in zip(tapes, hs):
            t.outs.append(h)
    return hs, tapes


def mlp_backward_group(tapes, d_outs, grads_list=None, need_input=False, workspace=None, plan=None, input_sink=None,
                       head_dx=None):
    """`mlp_backward` of G same-shaped tapes with grouped launches.  grads_list: [grads of tape g] or None; an entry may
    be None (no weight gradients for that tape).  need_input: one flag, or one per tape -- tapes that do not need
    d(input) drop out of the first layer's input-gradient launch (their slot of the result is None).
    input_sink(dys, ys, gate_act, ws): called INSTEAD of the first layer's input-gradient launch with the gradients at
    that layer's outputs, the outputs that gate them (or None) and the layer's weights, for the tapes that need d(input)
    -- a caller that only needs some columns of it (SAC's policy gradient: the action columns) computes them itself.
    head_dx: [gradient at the last hidden layer of tape g, already gated by its act'] -- replaces the last layer's
    input-gradient launch."""
    ds = list(d_outs)
    n = len(tapes[0].layers)
    want_in = list(need_input) if isinstance(need_input, (list, tuple)) else [bool(need_input)] * len(tapes)
    with_w = [] if grads_list is None else [g for g in range(len(tapes)) if grads_list[g] is not None]
    pregated = False                                  # ds already carries act' of layer k (see the one-output head below)
    for k in range(n - 1, -1, -1):
        last = k == n - 1
        gate_act = tapes[0].last_act if last else tapes[0].act
        gates = [None if (last and gate_act == _C.ACT_NONE) else t.outs[k] for t in tapes]
        if pregated:
            gates, gate_act, pregated = [None] * len(tapes), _C.ACT_NONE, False
        if with_w:
            inps = [(tapes[g].x if k == 0 else tapes[g].outs[k - 1]) for g in with_w]
            args = ([ds[g] for g in with_w], [gates[g] for g in with_w], gate_act, inps,
                    [grads_list[g][k][0] for g in with_w], [grads_list[g][k][1] for g in with_w])
            if plan is not None:
                _C.linear_bwd_weight_partials_group(*args, plan)
            else:
                _C.linear_bwd_weight_group(*args, workspace=workspace)
        if k > 0 and last and head_dx is not None:
            ds, pregated = list(head_dx), True
        elif k > 0 and last and gate_act == _C.ACT_NONE and int(tapes[0].layers[k][0].shape[0]) == 1 and \
                int(tapes[0].layers[k][0].shape[1]) % 4 == 0 and \
                all(t.outs[k - 1].data_ptr() % 16 == 0 for t in tapes):
            # a head with ONE output: d(hidden) = dq w^T is rank 1 -- produced already gated for the layer below by one
            # streaming launch, and that layer's GEMMs then read one operand instead of two
            ds = _C.outer_gate_group([d.reshape(-1) for d in ds], [t.layers[k][0].reshape(-1) for t in tapes],
                                     [t.outs[k - 1] for t in tapes], tapes[0].act)
            pregated = True
        elif k > 0:
            ds = _C.linear_bwd_input_group(ds, gates, gate_act, [t.layers[k][0] for t in tapes])
        elif any(want_in) and input_sink is not None:
            sel = [g for g in range(len(tapes)) if want_in[g]]
            input_sink([ds[g] for g in sel], [gates[g] for g in sel], gate_act, [tapes[g].layers[0][0] for g in sel])
            ds = [None] * len(tapes)
        elif any(want_in):
            sel = [g for g in range(len(tapes)) if want_in[g]]
            dx = _C.linear_bwd_input_group([ds[g] for g in sel], [gates[g] for g in sel], gate_act,
                                           [tapes[g].layers[0][0] for g in sel])
            ds = [None] * len(tapes)
            for g, d in zip(sel, dx):
                ds[g] = d
    if isinstance(need_input, (list, tuple)):
        return ds
    return ds if need_input else None


# ---------------------------------------------------------------------------------------------
# Conv trunks (CNNBase, torchrl/networks/base.py:59-107) + FC head, on im2col + the GEMM kernels.
# Activations are channels-last; the first layer reads uint8 NCHW frame stacks and scales them
# in-kernel (x / 255 - 0.5, ScaledFloatFrame).
class ConvTape:
    # feat_chw: the last conv layer's output (convs[-1][2]) is kept as (B, C, Ho*Wo) -- the flattened features themselves
    # dx_preps: {layer index: re-ordered weights of its implicit input-gradient kernel}, made by riders of the forward
    __slots__ = ("convs", "fc", "B", "act", "feat_shape", "feat_chw", "dx_preps")


def conv_layers(net):
    """[(Conv2d module), ...] of a Net whose trunk is a CNNBase without LayerNorm / padding."""
    mods = [m for m in net.base.seq_convs if isinstance(m, nn.Conv2d)]
    for m in mods:
        if tuple(m.padding) != (0, 0) or tuple(m.dilation) != (1, 1) or m.groups != 1:
            raise _C.TrlError("conv kernels support padding 0, dilation 1, groups 1 (got %s)" % m)
    if net.base.add_ln:
        raise _C.TrlError("conv kernels do not support LayerNorm")
    return mods


def fc_layers(net):
    return [(m.weight, m.bias) for m in net.seq_append_fcs if isinstance(m, nn.Linear)]


def cnn_act_code(net):
    a = net.base.activation_func
    if a not in ACT_OF or net.base.last_activation_func is not a:
        raise _C.TrlError("conv kernels support Tanh / ReLU (got %s)" % a)
    return ACT_OF[a]


def cnn_param_list(net):
    out = []
    for m in conv_layers(net):
        out += [m.weight, m.bias]
    for w, b in fc_layers(net):
        out += [w, b]
    return out


FWD_WEIGHT_SHADOWS = True       # (tests switch it off to compare)


PAIR_FIRST_CONV = True        # cnn_forward_pair: both networks' first conv layer as one launch (tests switch it off to compare)


def _shadow_jobs(layers):
    """[(layer index, weight matrix, C, kh*kw)] of the later conv layers that take the channels-last implicit GEMM: their
    weights are re-ordered to its (i, j, c) reduction order by riders of the first layer's launch (always from the live
    weights, no launch of their own), so that its B operand is dense 16-byte loads instead of strided 4-byte ones."""
    if not FWD_WEIGHT_SHADOWS:
        return []
    jobs = []
    for k in range(1, len(layers)):
        m = layers[k]
        if int(m.in_channels) % 4 == 0 and len(jobs) < 4:
            jobs.append((k, m.weight.view(m.weight.shape[0], -1), int(m.in_channels), int(m.kernel_size[0]) * int(m.kernel_size[1])))
    return jobs


def _dx_jobs(layers):
    """[(layer index, weight matrix, Cin, kh, kw, sh, sw)] of the conv layers whose input gradient is the implicit transposed
    convolution: their weight re-orderings ride on the first layer's forward launch when a backward pass will follow."""
    jobs = []
    for k in range(1, len(layers)):
        m = layers[k]
        geo = (int(m.in_channels), int(m.kernel_size[0]), int(m.kernel_size[1]), int(m.stride[0]), int(m.stride[1]))
        if _C.conv_bwd_input_ok(geo[0], int(m.weight.shape[0]), *geo[1:]) and len(jobs) < 4:
            jobs.append((k, m.weight.view(m.weight.shape[0], -1)) + geo)
    return jobs


def cnn_forward(net, frames_u8, scale=1.0 / 255.0, shift=-0.5, dx_prep=False, head=True):
    """frames_u8: (B, C, H, W) uint8.  Returns (out (B, O), tape).  dx_prep: a backward pass will follow -- the weight
    re-orderings of its input-gradient kernels ride on the first layer's launch (tape.dx_preps).  head=False: the pass
    stops at the last HIDDEN activations (the caller runs the linear head itself, e.g. `_C.dqn_act`)."""
    fcs = fc_layers(net)
    run_fc = (lambda feat: mlp_forward(fcs, feat, act)) if head else (lambda feat: mlp_forward(fcs[:-1], feat, act, last_act=act))
    if not head and len(fcs) < 2:
        raise _C.TrlError("cnn_forward(head=False) needs a hidden FC layer in front of the head")
    if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4:
        raise _C.TrlError("cnn_forward expects (B, C, H, W) uint8 frame stacks")
    act = cnn_act_code(net)
    t = ConvTape()
    t.convs, t.act, t.B, t.feat_chw, t.dx_preps = [], act, int(frames_u8.shape[0]), False, None
    x, geom_in = frames_u8.contiguous(), None
    layers = conv_layers(net)
    shadows = {}                                                             # layer index -> weight in the (i, j, c) reduction order
    for k, m in enumerate(layers):
        kh, kw = m.kernel_size
        sh, sw = m.stride
        if k == 0 and _C.conv_u8_implicit_ok(x, kh, kw, sh, sw):
            # implicit GEMM straight from the uint8 frames: the im2col matrix (210 MB at cfg 5) never exists
            wmat = m.weight.view(m.weight.shape[0], -1)
            jobs, dxj = _shadow_jobs(layers), (_dx_jobs(layers) if dx_prep else [])
            y, (B, Ho, Wo), (outs, wss) = _C.conv_fwd_u8(x, wmat, m.bias, kh, kw, sh, sw, scale, shift, act,
                                                         perm=[j[1:] for j in jobs], dx=[j[1:] for j in dxj])
            shadows = {j[0]: o for j, o in zip(jobs, outs)}
            t.dx_preps = {j[0]: ws for j, ws in zip(dxj, wss)}
            t.convs.append(("u8", (x, scale, shift), y, wmat, None, (kh, kw, sh, sw)))
            x = y.view(B, Ho, Wo, int(wmat.shape[0]))
            continue
        if k > 0 and int(x.shape[3]) % 4 == 0:
            # implicit GEMM on the channels-last activations (reduction in (i, j, c) order: contiguous window rows)
            # (the last layer stores its output in nn.Flatten's (c, oy, ox) order: no transposing launch in front of the FCs)
            wmat = m.weight.view(m.weight.shape[0], -1)
            in_shape = tuple(int(v) for v in x.shape)
            t.feat_chw = k == len(layers) - 1
            y, (B, Ho, Wo) = _C.conv_fwd_nhwc(x, shadows.get(k, wmat), m.bias, kh, kw, sh, sw, act, out_chw=t.feat_chw,
                                              w_perm=k in shadows)
            t.convs.append(("nhwc", x, y, wmat, in_shape, (kh, kw, sh, sw)))
            if t.feat_chw:
                t.feat_shape = (Ho * Wo, int(wmat.shape[0]))
                out, t.fc = run_fc(y)
                return out, t
            x = y.view(B, Ho, Wo, int(wmat.shape[0]))
            continue
        if k == 0:
            cols, (B, Ho, Wo) = _C.im2col(x, kh, kw, sh, sw, scale=scale, shift=shift)
            in_shape = None
        else:
            in_shape = tuple(int(v) for v in x.shape)                        # (B, H, W, C)
            cols, (B, Ho, Wo) = _C.im2col(x, kh, kw, sh, sw)
        wmat = m.weight.view(m.weight.shape[0], -1)
        y = _C.linear_fwd(cols, wmat, m.bias, act)                           # (B*Ho*Wo, Cout) == NHWC
        t.convs.append(("im2col", cols, y, wmat, in_shape, (kh, kw, sh, sw)))
        x = y.view(B, Ho, Wo, int(wmat.shape[0]))
    B, Ho, Wo, Cc = (int(v) for v in x.shape)
    t.feat_shape = (Ho * Wo, Cc)
    feat = _C.transpose_bpc(x, B, Ho * Wo, Cc).view(B, Cc * Ho * Wo)       # PyTorch's NCHW flatten order
    out, t.fc = run_fc(feat)
    return out, t


def cnn_forward_pair(net_a, net_b, frames_a, frames_b, scale=1.0 / 255.0, shift=-0.5, head=True, dx_prep=False):
    """`cnn_forward` of two same-architecture networks on two frame batches -- DQN's online net on obs and target net on
    next_obs -- with every layer after the first as ONE grouped launch (conv 2 / 3 as grouped implicit GEMMs, the FC head
    as grouped (split-K) layers).  Returns ((out_a, tape_a), (out_b, tape_b)); falls back to two separate passes for
    geometries outside the grouped paths.  Same kernels and arithmetic as `cnn_forward`.
    head=False: the pass stops at the last HIDDEN activations (the caller runs the linear head itself, `_C.dqn_head`);
    the tapes then end at that layer and `cnn_backward` takes the gradient w.r.t. those activations.  Returns None when
    the grouped path does not apply (the caller then runs the full pass)."""
    convs_a, convs_b = conv_layers(net_a), conv_layers(net_b)
    act = cnn_act_code(net_a)
    same = (len(convs_a) == len(convs_b) and cnn_act_code(net_b) == act and tuple(frames_a.shape) == tuple(frames_b.shape)
            and all(ma.weight.shape == mb.weight.shape and ma.kernel_size == mb.kernel_size and ma.stride == mb.stride
                    for ma, mb in zip(convs_a, convs_b))
            and [tuple(w.shape) for w, _ in fc_layers(net_a)] == [tuple(w.shape) for w, _ in fc_layers(net_b)])
    first = convs_a[0] if convs_a else None
    if not same or first is None or frames_a.dtype != torch.uint8 or \
            not _C.conv_u8_implicit_ok(frames_a, *first.kernel_size, *first.stride) or \
            any(int(m.in_channels) % 4 for m in convs_a[1:]):
        if not head:
            return None
        return cnn_forward(net_a, frames_a, scale, shift, dx_prep=dx_prep), cnn_forward(net_b, frames_b, scale, shift)
    if not head and len(fc_layers(net_a)) < 2:
        return None
    tapes, xs, shadows = [], [], []
    m = convs_a[0]
    kh, kw = m.kernel_size
    sh, sw = m.stride
    jobs_a, jobs_b = _shadow_jobs(convs_a), _shadow_jobs(convs_b)
    dxj = _dx_jobs(convs_a) if dx_prep else []                            # (net_a's backward follows)
    paired = PAIR_FIRST_CONV and len(jobs_a) + len(jobs_b) <= 4 and (convs_a[0].bias is None) == (convs_b[0].bias is None)
    if paired:
        # both first layers -- online net on obs, target net on next_obs -- as ONE launch that also carries both networks'
        # weight re-orderings (trl_conv_fwd_u8_pair_f32)
        wa, wb = (c[0].weight.view(c[0].weight.shape[0], -1) for c in (convs_a, convs_b))
        fa, fb = frames_a.contiguous(), frames_b.contiguous()
        ya, yb, (B, Ho, Wo), (outs, wss) = _C.conv_fwd_u8_pair(fa, wa, convs_a[0].bias, fb, wb, convs_b[0].bias, kh, kw, sh, sw,
                                                               scale, shift, act, perm=[j[1:] for j in jobs_a + jobs_b],
                                                               dx=[j[1:] for j in dxj])
        first = ((fa, wa, ya, jobs_a, outs[:len(jobs_a)], dxj, wss), (fb, wb, yb, jobs_b, outs[len(jobs_a):], [], []))
    for idx, (net, convs, frames) in enumerate(((net_a, convs_a, frames_a), (net_b, convs_b, frames_b))):
        t = ConvTape()
        t.convs, t.act, t.B, t.feat_chw, t.dx_preps = [], act, int(frames.shape[0]), False, None
        if paired:
            fr, wmat, y, jobs, outs_, dxj_, wss_ = first[idx]
        else:
            wmat = convs[0].weight.view(convs[0].weight.shape[0], -1)
            fr = frames.contiguous()
            jobs, dxj_ = (jobs_a, dxj) if net is net_a else (jobs_b, [])
            y, (B, Ho, Wo), (outs_, wss_) = _C.conv_fwd_u8(fr, wmat, convs[0].bias, kh, kw, sh, sw, scale, shift, act,
                                                           perm=[j[1:] for j in jobs], dx=[j[1:] for j in dxj_])
        shadows.append({j[0]: o for j, o in zip(jobs, outs_)})
        t.dx_preps = {j[0]: ws for j, ws in zip(dxj_, wss_)}
        t.convs.append(("u8", (fr, scale, shift), y, wmat, None, (kh, kw, sh, sw)))
        tapes.append(t)
        xs.append(y.view(B, Ho, Wo, int(wmat.shape[0])))
    for k in range(1, len(convs_a)):
        ma, mb = convs_a[k], convs_b[k]
        kh, kw = ma.kernel_size
        sh, sw = ma.stride
        wmats = [m.weight.view(m.weight.shape[0], -1) for m in (ma, mb)]
        in_shape = tuple(int(v) for v in xs[0].shape)
        last = k == len(convs_a) - 1
        perm = all(k in sh_ for sh_ in shadows)
        ys, (B, Ho, Wo) = _C.conv_fwd_nhwc_group(xs, [sh_[k] for sh_ in shadows] if perm else wmats, [ma.bias, mb.bias],
                                                 kh, kw, sh, sw, act, out_chw=last, w_perm=perm)
        for t, x, y, wmat in zip(tapes, xs, ys, wmats):
            t.convs.append(("nhwc", x, y, wmat, in_shape, (kh, kw, sh, sw)))
            t.feat_chw = last
        if last:
            feats = ys
            for t in tapes:
                t.feat_shape = (Ho * Wo, int(wmats[0].shape[0]))
            break
        xs = [y.view(B, Ho, Wo, int(wmats[0].shape[0])) for y in ys]
    else:                                                      # a one-layer trunk: features of the uint8 layer
        B, Ho, Wo, Cc = (int(v) for v in xs[0].shape)
        feats = []
        for t, x in zip(tapes, xs):
            t.feat_shape = (Ho * Wo, Cc)
            feats.append(_C.transpose_bpc(x, B, Ho * Wo, Cc).view(B, Cc * Ho * Wo))   # PyTorch's NCHW flatten order
    if head:
        outs, fcs = mlp_forward_group([fc_layers(net_a), fc_layers(net_b)], feats, act)
    else:
        outs, fcs = mlp_forward_group([fc_layers(net_a)[:-1], fc_layers(net_b)[:-1]], feats, act, last_act=act)
    for t, fc in zip(tapes, fcs):
        t.fc = fc
    return (outs[0], tapes[0]), (outs[1], tapes[1])


def _conv_ws_needs(tape):
    """floats of split-partial workspace of every conv layer's weight gradient (each layer its own region: their folds
    run together at the end of the trunk, _C.FoldScope)"""
    needs = []
    for kind, src, y, wmat, in_shape, (kh, kw, sh, sw) in tape.convs:
        Cout = int(wmat.shape[0])
        if kind == "nhwc":
            B, H, W, Cc = (int(v) for v in src.shape)
            n = _C.lib().trl_conv_bwd_weight_workspace(B, Cc, H, W, kh, kw, sh, sw, Cout)
        elif kind == "u8":
            B, Cc, H, W = (int(v) for v in src[0].shape)
            n = _C.lib().trl_conv_bwd_weight_workspace(B, Cc, H, W, kh, kw, sh, sw, Cout)
        else:
            n = _C.lib().trl_linear_bwd_weight_workspace(y.numel() // Cout, int(wmat.shape[1]), Cout)
        if n < 0:
            raise _C.TrlError("cnn_backward: bad conv geometry")
        needs.append((int(n) + 3) & ~3)                                    # regions stay 16-byte aligned
    return needs


def cnn_backward_workspace(tape):
    return sum(_conv_ws_needs(tape))


def cnn_backward(net, tape, d_out, grads, workspace=None):
    """grads: [(dW_view, db_view), ...] in cnn_param_list order (conv layers, then FC layers).
    The gradient flowing down the trunk is gated ONCE, where it is produced (dZ = dY * act'(Y): in the transpose that
    un-flattens d(features), and in the epilogue of each implicit input-gradient kernel), so the weight- and
    input-gradient kernels of a layer read one tensor instead of two.  The conv layers' split weight-gradient partials are
    folded by ONE launch at the end of the trunk (_C.FoldScope) instead of one per layer."""
    n_conv = len(tape.convs)
    d_feat = mlp_backward(tape.fc, d_out, grads=grads[n_conv:], need_input=True, workspace=workspace)
    P, Cc = tape.feat_shape
    top_y = tape.convs[-1][2]
    d = _C.transpose_bpc(d_feat.view(tape.B, Cc, P), tape.B, Cc, P, y_gate=top_y, gate_act=tape.act,
                         gate_like_in=tape.feat_chw).view(tape.B * P, Cc)   # back to (B, P, C)
    gated = True                                                         # d is dZ of layer k (else dY)
    needs = _conv_ws_needs(tape)
    if workspace is None or workspace.numel() < sum(needs) or workspace.data_ptr() % 16:
        workspace = torch.empty(sum(needs), dtype=torch.float32, device=d.device)
    offs = [sum(needs[:k]) for k in range(n_conv)]
    with _C.FoldScope(d.device):
        d = _cnn_trunk_backward(tape, d, grads, gated, [workspace[o:o + n] for o, n in zip(offs, needs)])


def _cnn_trunk_backward(tape, d, grads, gated, regions):
    n_conv = len(tape.convs)
    # the implicit input-gradient kernels read the weights re-ordered: all layers' re-orderings in ONE launch up front
    dx_layers = [k for k in range(1, n_conv)
                 if _C.conv_bwd_input_ok(tape.convs[k][4][3], int(tape.convs[k][3].shape[0]), *tape.convs[k][5])]
    preps = dict(getattr(tape, "dx_preps", None) or {})                     # (made by riders of the forward's first launch)
    todo = [k for k in dx_layers if k not in preps]
    if len(todo) > 1:
        wss = _C.conv_bwd_input_prep([(tape.convs[k][3], tape.convs[k][4][3]) + tuple(tape.convs[k][5]) for k in todo], d.device)
        preps.update(zip(todo, wss))
    for k in range(n_conv - 1, -1, -1):
        kind, src, y, wmat, in_shape, (kh, kw, sh, sw) = tape.convs[k]      # src: cols matrix / input activations / frames
        gw, gb = grads[k]
        workspace = regions[k]
        yg, ga = (None, _C.ACT_NONE) if gated else (y, tape.act)
        if kind == "nhwc":                                                   # implicit GEMM on channels-last activations
            _C.conv_bwd_weight_nhwc(d, yg, ga, src, kh, kw, sh, sw, gw.view(wmat.shape), gb, workspace=workspace)
        elif kind == "u8":                                                   # first layer, implicit GEMM on the frames
            frames, scale, shift = src
            _C.conv_bwd_weight_u8(d, yg, ga, frames, kh, kw, sh, sw, scale, shift, gw.view(wmat.shape), gb,
                                  workspace=workspace)
        else:
            _C.linear_bwd_weight(d, yg, ga, src, dw=gw.view(wmat.shape), db=gb, workspace=workspace)
        if k > 0:
            B, H, W, Cin = in_shape
            if _C.conv_bwd_input_ok(Cin, int(wmat.shape[0]), kh, kw, sh, sw):
                # implicit transposed convolution: no (B Ho Wo) x (Cin kh kw) matrix in between; its epilogue applies
                # the NEXT layer down's act'
                below = tape.convs[k - 1][2]
                d = _C.conv_bwd_input_nhwc(d, yg, ga, wmat, B, Cin, H, W, kh, kw, sh, sw, x_gate=below,
                                           x_gate_act=tape.act, prep=preps.get(k)).view(B * H * W, Cin)
                gated = True
            else:
                dcols = _C.linear_bwd_input(d, yg, ga, wmat)
                d = _C.col2im(dcols, B, Cin, H, W, kh, kw, sh, sw).view(B * H * W, Cin)
                gated = False
