"""DQN / QR-DQN path on the GPU: conv layers (im2col + MFMA GEMM) forward / backward vs torch,
TD and quantile-Huber loss kernels and whole updates vs the reference's outputs
(tests/golden/dqn.npz), epsilon-greedy policy and frame env vs the CPU oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.dqn import SynthFrameVecEnvCPU, cnn, scale_frames

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
CONVS = [[8, [8, 8], [4, 4], [0, 0]], [8, [4, 4], [2, 2], [0, 0]], [16, [3, 3], [1, 1], [0, 0]]]


def small_qnet(A, Q=1, act=torch.nn.Tanh):
    import torchrl.networks as networks
    return networks.Net(output_shape=A * Q, base_type=networks.CNNBase, append_hidden_shapes=[32],
                        activation_func=act, input_shape=(4, 84, 84), hidden_shapes=CONVS)


@pytest.mark.parametrize("act", [torch.nn.Tanh, torch.nn.ReLU])
def test_cnn_forward_backward_vs_torch(act):
    from torchrl_amd import ops
    torch.manual_seed(1)
    net = small_qnet(6, act=act)
    frames = torch.randint(0, 256, (5, 4, 84, 84), dtype=torch.uint8)
    params = [p.detach().clone().requires_grad_(True) for p in ops.cnn_param_list(net)]
    want = cnn(scale_frames(frames.numpy()), params, [4, 2, 1], "tanh" if act is torch.nn.Tanh else "relu")
    d_out = torch.randn(5, 6)
    want.backward(d_out)
    net.to(DEV)
    plist = ops.cnn_param_list(net)
    out, tape = ops.cnn_forward(net, frames.to(DEV))
    assert (out.cpu() - want.detach()).abs().max().item() < 2e-5
    grads = [(torch.zeros_like(plist[k]), torch.zeros_like(plist[k + 1])) for k in range(0, len(plist), 2)]
    ops.cnn_backward(net, tape, d_out.to(DEV), grads)
    flat = [t for pair in grads for t in pair]
    for g, p in zip(flat, params):
        err = (g.cpu() - p.grad).abs().max().item()
        assert err < 5e-5 * max(1.0, p.grad.abs().max().item()), err


@pytest.mark.parametrize("B,C,H,W,kh,kw,sh,sw,Cout", [
    (5, 4, 84, 84, 8, 8, 4, 4, 16),        # the Atari first layer (dqn_pong.json): direct register-weights kernels
    (3, 1, 36, 44, 8, 8, 4, 4, 5),         # direct kernels, K = 64, 5 channels, M = 240 (ragged wave)
    (9, 2, 40, 36, 8, 8, 4, 4, 16),        # direct kernels, K = 128, M = 648
    (4, 3, 33, 44, 8, 8, 3, 4, 12),        # direct kernels, K = 192, vertical stride 3
    (70, 4, 84, 84, 8, 8, 4, 4, 16),       # direct kernels, M = 28 000: many workgroups in the weight gradient
    (3, 3, 36, 44, 3, 4, 2, 4, 5),         # K = 36 (one partial panel), ragged M, Cout < tile
    (2, 2, 20, 20, 8, 8, 4, 4, 70),        # Ho * Wo = 16 (power-of-two divisor), two N tiles
    (7, 1, 9, 8, 2, 8, 1, 4, 3),           # Wo = 1 (division by 1), Ho = 8
    (130, 5, 16, 16, 4, 4, 4, 4, 33),      # M = 2080: several reduction splits in the weight gradient
])
@pytest.mark.parametrize("act", ["relu", "tanh"])
def test_implicit_gemm_first_conv_layer_vs_torch(B, C, H, W, kh, kw, sh, sw, Cout, act):
    """trl_conv_fwd_u8_f32 / trl_conv_bwd_weight_u8_f32 (no im2col buffer) against torch's conv2d + autograd on
    the scaled frames."""
    from torchrl_amd import _C
    gen = torch.Generator().manual_seed(B * 1000 + Cout)
    frames = torch.randint(0, 256, (B, C, H, W), dtype=torch.uint8, generator=gen)
    w = (torch.randn(Cout, C, kh, kw, generator=gen) / (C * kh * kw) ** 0.5).requires_grad_(True)
    b = torch.randn(Cout, generator=gen).requires_grad_(True)
    f = {"relu": torch.relu, "tanh": torch.tanh}[act]
    code = {"relu": _C.ACT_RELU, "tanh": _C.ACT_TANH}[act]
    want = f(F.conv2d(frames.float() / 255.0 - 0.5, w, b, stride=(sh, sw)))           # (B, Cout, Ho, Wo)
    dy = torch.randn(want.shape, generator=gen)
    want.backward(dy)
    assert _C.conv_u8_implicit_ok(frames, kh, kw, sh, sw)
    wd, bd = w.detach().to(DEV), b.detach().to(DEV)
    y, (Bo, Ho, Wo) = _C.conv_fwd_u8(frames.to(DEV), wd.view(Cout, -1), bd, kh, kw, sh, sw, 1.0 / 255.0, -0.5, code)
    got = y.view(B, Ho, Wo, Cout).permute(0, 3, 1, 2).cpu()
    assert (got - want.detach()).abs().max().item() < 2e-5
    dy_rows = dy.permute(0, 2, 3, 1).reshape(-1, Cout).contiguous().to(DEV)
    dw, db = torch.empty_like(wd.view(Cout, -1)), torch.empty_like(bd)
    _C.conv_bwd_weight_u8(dy_rows, y, code, frames.to(DEV), kh, kw, sh, sw, 1.0 / 255.0, -0.5, dw, db)
    scale = lambda t: max(1.0, t.abs().max().item())
    assert (dw.cpu().view_as(w) - w.grad).abs().max().item() < 5e-5 * scale(w.grad)
    assert (db.cpu() - b.grad).abs().max().item() < 5e-5 * scale(b.grad)


@pytest.mark.parametrize("B,C,H,W,kh,kw,sh,sw,Cout", [
    (6, 16, 20, 20, 4, 4, 2, 2, 32),       # second Atari layer
    (6, 32, 9, 9, 3, 3, 1, 1, 64),         # third Atari layer: K = 288 (2.25 panels)
    (3, 8, 11, 13, 3, 2, 2, 3, 5),         # K = 48, ragged everything, Cout < tile
    (130, 4, 6, 6, 3, 3, 1, 1, 70),        # M = 2080: reduction splits in the weight gradient; two N tiles
    (2, 12, 5, 4, 5, 4, 1, 1, 7),          # Ho = Wo = 1
])
@pytest.mark.parametrize("act", ["relu", "tanh"])
def test_implicit_gemm_channels_last_conv_vs_torch(B, C, H, W, kh, kw, sh, sw, Cout, act):
    """trl_conv_fwd_nhwc_f32 / trl_conv_bwd_weight_nhwc_f32: reduction in (i, j, c) order over (B, H, W, C)
    activations, nn.Conv2d weight layout in and out, against torch's conv2d + autograd."""
    from torchrl_amd import _C
    gen = torch.Generator().manual_seed(B * 100 + Cout)
    x = torch.randn(B, C, H, W, generator=gen)
    w = (torch.randn(Cout, C, kh, kw, generator=gen) / (C * kh * kw) ** 0.5).requires_grad_(True)
    b = torch.randn(Cout, generator=gen).requires_grad_(True)
    f = {"relu": torch.relu, "tanh": torch.tanh}[act]
    code = {"relu": _C.ACT_RELU, "tanh": _C.ACT_TANH}[act]
    want = f(F.conv2d(x, w, b, stride=(sh, sw)))
    dy = torch.randn(want.shape, generator=gen)
    want.backward(dy)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wd, bd = w.detach().to(DEV), b.detach().to(DEV)
    y, (Bo, Ho, Wo) = _C.conv_fwd_nhwc(x_nhwc, wd.view(Cout, -1), bd, kh, kw, sh, sw, code)
    got = y.view(B, Ho, Wo, Cout).permute(0, 3, 1, 2).cpu()
    assert (got - want.detach()).abs().max().item() < 2e-5
    dy_rows = dy.permute(0, 2, 3, 1).reshape(-1, Cout).contiguous().to(DEV)
    dw, db = torch.empty_like(wd.view(Cout, -1)), torch.empty_like(bd)
    _C.conv_bwd_weight_nhwc(dy_rows, y, code, x_nhwc, kh, kw, sh, sw, dw, db)
    scale = lambda t: max(1.0, t.abs().max().item())
    assert (dw.cpu().view_as(w) - w.grad).abs().max().item() < 5e-5 * scale(w.grad)
    assert (db.cpu() - b.grad).abs().max().item() < 5e-5 * scale(b.grad)


@pytest.mark.parametrize("B,C,H,W,kh,kw,sh,sw,Cout", [
    (512, 16, 20, 20, 4, 4, 2, 2, 32),     # cfg 5 conv 2 at full size: four parity classes of 51 200 positions
    (512, 32, 9, 9, 3, 3, 1, 1, 64),       # cfg 5 conv 3: one class, nine taps, 74 KB of weights in LDS
    (3, 16, 11, 13, 3, 3, 2, 2, 16),       # (H - kh) % sh != 0: trailing rows no window covers (gradient 0); classes
    (2, 48, 9, 8, 5, 4, 3, 2, 32),         #   with different tap counts; three column blocks
    (5, 64, 6, 6, 3, 3, 1, 1, 16),         # four column blocks; 180 positions: partial last tile
    (2, 16, 5, 4, 5, 4, 1, 1, 16),         # Ho = Wo = 1
])
@pytest.mark.parametrize("act", ["relu", "tanh", "none"])
@pytest.mark.parametrize("form", ["image", "class"])
def test_implicit_transposed_conv_input_gradient_vs_torch(B, C, H, W, kh, kw, sh, sw, Cout, act, form, monkeypatch):
    """trl_conv_bwd_input_nhwc_f32 (no cols matrix) against autograd's conv2d input gradient, and against the
    cols-GEMM + col2im pair it replaces.  Both forms of the kernel: `image` (the gated dZ of whole images staged in LDS once;
    what every geometry whose weights + one image fit in LDS runs) and `class` (dZ re-read per tap: the larger layers),
    which must agree bit for bit -- the same MFMA sequence per output element."""
    from torchrl_amd import _C
    monkeypatch.setenv("TRL_DX_CLASS_FORM", "1" if form == "class" else "0")
    gen = torch.Generator().manual_seed(B * 10 + C + Cout)
    x = torch.randn(B, C, H, W, generator=gen).requires_grad_(True)
    w = torch.randn(Cout, C, kh, kw, generator=gen) / (C * kh * kw) ** 0.5
    f = {"relu": torch.relu, "tanh": torch.tanh, "none": lambda t: t}[act]
    code = {"relu": _C.ACT_RELU, "tanh": _C.ACT_TANH, "none": _C.ACT_NONE}[act]
    y = f(F.conv2d(x, w, None, stride=(sh, sw)))
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy)
    assert _C.conv_bwd_input_ok(C, Cout, kh, kw, sh, sw)
    rows = lambda t: t.detach().permute(0, 2, 3, 1).reshape(-1, Cout).contiguous().to(DEV)
    wd = w.view(Cout, -1).to(DEV)
    gate = rows(y) if act != "none" else None
    got = _C.conv_bwd_input_nhwc(rows(dy), gate, code, wd, B, C, H, W, kh, kw, sh, sw)
    want = x.grad.permute(0, 2, 3, 1)
    scale = max(1.0, want.abs().max().item())
    assert (got.cpu() - want).abs().max().item() < 2e-5 * scale
    old = _C.col2im(_C.linear_bwd_input(rows(dy), gate, code, wd), B, C, H, W, kh, kw, sh, sw)
    assert (got - old).abs().max().item() < 2e-5 * scale
    if form == "image":
        monkeypatch.setenv("TRL_DX_CLASS_FORM", "1")
        assert torch.equal(_C.conv_bwd_input_nhwc(rows(dy), gate, code, wd, B, C, H, W, kh, kw, sh, sw), got)
        monkeypatch.setenv("TRL_DX_CLASS_FORM", "0")
        for img in ("1", "3"):                                              # images per workgroup: ragged last group
            monkeypatch.setenv("TRL_DX_IMG", img)
            assert torch.equal(_C.conv_bwd_input_nhwc(rows(dy), gate, code, wd, B, C, H, W, kh, kw, sh, sw), got)
        monkeypatch.delenv("TRL_DX_IMG")
    # the weights re-ordered ahead, together with another layer's, by ONE launch (trl_conv_bwd_input_nhwc_prep_f32)
    other = torch.randn(32, 16 * 3 * 3, generator=torch.Generator().manual_seed(1)).to(DEV)   # (its own stream: `gen` goes on below)
    preps = _C.conv_bwd_input_prep([(other, 16, 3, 3, 1, 1), (wd, C, kh, kw, sh, sw)], DEV)
    assert torch.equal(_C.conv_bwd_input_nhwc(rows(dy), gate, code, wd, B, C, H, W, kh, kw, sh, sw, prep=preps[1]), got)
    # epilogue gate: the layer below receives dZ = dX * act'(its own output)
    below = torch.tanh(torch.randn(B, H, W, C, generator=gen)).to(DEV)
    gated = _C.conv_bwd_input_nhwc(rows(dy), gate, code, wd, B, C, H, W, kh, kw, sh, sw, x_gate=below.view(-1, C),
                                   x_gate_act=_C.ACT_TANH)
    assert torch.allclose(gated, got * (1.0 - below * below), rtol=1e-6, atol=1e-7)   # (1 - y y) contracts to an fma
    # ... and the un-flattening transpose gates the same way
    feat = torch.randn(B, C, H * W, generator=gen).to(DEV)
    yb = below.view(B, H * W, C)
    assert torch.allclose(_C.transpose_bpc(feat, B, C, H * W, y_gate=yb, gate_act=_C.ACT_TANH),
                          _C.transpose_bpc(feat, B, C, H * W) * (1.0 - yb * yb), rtol=1e-6, atol=1e-7)


def test_implicit_transposed_conv_coverage_and_errors():
    from torchrl_amd import _C
    assert not _C.conv_bwd_input_ok(4, 16, 8, 8, 4, 4) and not _C.conv_bwd_input_ok(16, 24, 3, 3, 1, 1)
    assert not _C.conv_bwd_input_ok(128, 16, 3, 3, 1, 1) and not _C.conv_bwd_input_ok(16, 16, 2, 2, 3, 3)
    assert not _C.conv_bwd_input_ok(64, 64, 5, 5, 1, 1)                      # 25 taps x 64 x 64 floats > LDS
    with pytest.raises(_C.TrlError, match="multiple of 16"):
        _C.conv_bwd_input_nhwc(torch.zeros(8, 16, device=DEV), None, _C.ACT_NONE, torch.zeros(16, 4 * 9, device=DEV),
                               2, 4, 4, 4, 3, 3, 1, 1)


@pytest.mark.parametrize("B,dx_prep,pair_first", [(96, False, True), (37, True, True), (96, True, False)])
def test_paired_forward_of_online_and_target_net_equals_two_passes(B, dx_prep, pair_first, monkeypatch):
    """ops.cnn_forward_pair (both first conv layers as ONE launch -- trl_conv_fwd_u8_pair_f32, with both networks' weight
    re-orderings as its riders --, conv 2 / 3, the FC layer and the head of both networks as grouped launches) against two
    ops.cnn_forward passes: outputs and every tape tensor bit for bit, at cfg 5's shapes; a ragged batch (37 x 400 positions:
    the second problem's workgroups start mid-way through a round) and the two-launch first layer as well."""
    import copy
    import torchrl.networks as networks
    from torchrl_amd import ops
    monkeypatch.setattr(ops, "PAIR_FIRST_CONV", pair_first)
    torch.manual_seed(2)
    convs = [[16, [8, 8], [4, 4], [0, 0]], [32, [4, 4], [2, 2], [0, 0]], [64, [3, 3], [1, 1], [0, 0]]]
    qf = networks.Net(output_shape=6, base_type=networks.CNNBase, append_hidden_shapes=[512], activation_func=torch.nn.Tanh,
                      input_shape=(4, 84, 84), hidden_shapes=convs).to(DEV)
    tq = copy.deepcopy(qf)
    with torch.no_grad():
        for p in tq.parameters():
            p.add_(0.01 * torch.randn_like(p))
    fa = torch.randint(0, 256, (B, 4, 84, 84), dtype=torch.uint8, device=DEV)
    fb = torch.randint(0, 256, (B, 4, 84, 84), dtype=torch.uint8, device=DEV)
    (qa, ta), (qb, tb) = ops.cnn_forward_pair(qf, tq, fa, fb, dx_prep=dx_prep)
    wa, wta = ops.cnn_forward(qf, fa, dx_prep=dx_prep)
    wb, wtb = ops.cnn_forward(tq, fb)
    assert torch.equal(qa, wa) and torch.equal(qb, wb)
    if dx_prep:                                                           # the backward pass's re-ordered weights rode along
        assert sorted(ta.dx_preps) == sorted(wta.dx_preps) and all(torch.equal(ta.dx_preps[k], wta.dx_preps[k]) for k in ta.dx_preps)
        assert not tb.dx_preps
    for got, want in ((ta, wta), (tb, wtb)):
        assert len(got.convs) == len(want.convs) == 3 and got.feat_shape == want.feat_shape
        for g, w in zip(got.convs, want.convs):
            assert g[0] == w[0] and torch.equal(g[2], w[2]) and g[4] == w[4] and g[5] == w[5]
        for g, w in zip(got.fc.outs, want.fc.outs):
            assert torch.equal(g, w)
    # the backward pass runs on a tape of either origin
    d = torch.randn_like(qa)
    gp = [torch.zeros_like(p) for p in ops.cnn_param_list(qf)]
    gw = [torch.zeros_like(p) for p in ops.cnn_param_list(qf)]
    pair = lambda g: [(g[k], g[k + 1]) for k in range(0, len(g), 2)]
    ops.cnn_backward(qf, ta, d, pair(gp))
    ops.cnn_backward(qf, wta, d, pair(gw))
    for x, y in zip(gp, gw):
        assert torch.equal(x, y)


def test_implicit_gemm_rejects_unaligned_geometry():
    from torchrl_amd import _C
    frames = torch.zeros(2, 4, 21, 21, dtype=torch.uint8, device=DEV)
    assert not _C.conv_u8_implicit_ok(frames, 3, 3, 2, 2)
    with pytest.raises(_C.TrlError, match="multiples of 4"):
        _C.conv_fwd_u8(frames, torch.zeros(8, 36, device=DEV), None, 3, 3, 2, 2, 1.0, 0.0, _C.ACT_RELU)
    with pytest.raises(_C.TrlError, match="C % 4"):
        _C.conv_fwd_nhwc(torch.zeros(2, 8, 8, 6, device=DEV), torch.zeros(8, 54, device=DEV), None, 3, 3, 1, 1, _C.ACT_RELU)


def test_im2col_col2im_transpose_vs_torch():
    from torchrl_amd import _C
    x = torch.randn(3, 9, 11, 5)                                             # (B, H, W, C)
    cols, (B, Ho, Wo) = _C.im2col(x.to(DEV), 4, 3, 2, 2)
    want = F.unfold(x.permute(0, 3, 1, 2), (4, 3), stride=(2, 2))            # (B, C*kh*kw, L)
    assert torch.equal(cols.cpu().view(B, Ho * Wo, -1), want.transpose(1, 2).contiguous())
    d = torch.randn_like(cols.cpu())
    dx = _C.col2im(d.to(DEV), 3, 5, 9, 11, 4, 3, 2, 2)
    want_dx = F.fold(d.view(B, Ho * Wo, -1).transpose(1, 2), (9, 11), (4, 3), stride=(2, 2)).permute(0, 2, 3, 1)
    assert (dx.cpu() - want_dx).abs().max().item() < 1e-5
    u8 = torch.randint(0, 256, (2, 4, 20, 20), dtype=torch.uint8)
    c8, _ = _C.im2col(u8.to(DEV), 8, 8, 4, 4, scale=1 / 255.0, shift=-0.5)
    w8 = F.unfold(u8.float() / 255.0 - 0.5, 8, stride=4).transpose(1, 2)
    assert (c8.cpu().view(2, -1, 256) - w8).abs().max().item() < 1e-6
    t = torch.randn(4, 7, 3)
    assert torch.equal(_C.transpose_bpc(t.to(DEV), 4, 7, 3).cpu(), t.transpose(1, 2).contiguous())


def test_quantile_huber_kernel_vs_reference(golden):
    from torchrl_amd import _C
    g = golden("dqn")
    src, tgt = torch.tensor(g["qr_src"]), torch.tensor(g["qr_tgt"])
    B, Q = src.shape
    # one action, gamma 1, rewards 0, not terminal: target == next quantiles, theta == src
    sums = torch.zeros(3, dtype=torch.float64, device=DEV)
    dq = _C.quantile_huber(src.to(DEV).contiguous(), torch.zeros(B, dtype=torch.int64, device=DEV), tgt.to(DEV).contiguous(),
                           torch.zeros(B, device=DEV), torch.zeros(B, device=DEV), 1.0, 1, Q, sums)
    assert abs(sums[0].item() / (B * Q * Q) - float(g["qr_loss"])) < 1e-6
    np.testing.assert_allclose(dq.cpu().numpy(), g["qr_grad"], atol=2e-9, rtol=1e-5)


def dqn_batches(g, tag):
    B, Q, A, steps, seed = (int(x) for x in g[f"{tag}_args"])
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(steps):
        obs = rs.randint(0, 256, size=(B, 4, 84, 84)).astype(np.uint8)
        nobs = rs.randint(0, 256, size=(B, 4, 84, 84)).astype(np.uint8)
        acts = rs.randint(0, A, size=(B, 1) if Q == 1 else (B,))
        out.append({"obs": torch.from_numpy(obs), "next_obs": torch.from_numpy(nobs), "acts": acts,
                    "rewards": rs.randn(B, 1).astype(np.float32), "terminals": (rs.rand(B, 1) < 0.2).astype(np.float32)})
    return out


@pytest.mark.parametrize("tag", ["dqn", "qrdqn"])
def test_dqn_updates_match_reference(golden, tag):
    from torchrl.algo import DQN, QRDQN
    from torchrl.env.synth import SynthFrameVecEnv
    from torchrl.policies import EpsilonGreedyDQNDiscretePolicy
    g = golden("dqn")
    B, Q, A, steps, _ = (int(x) for x in g[f"{tag}_args"])
    qf = small_qnet(A, Q)
    qf.load_state_dict({k[len(tag) + 5:].replace("__", "."): torch.tensor(g[k]) for k in g.files if k.startswith(f"{tag}_qf0_")})
    env = SynthFrameVecEnv(4, device=DEV)
    pf = EpsilonGreedyDQNDiscretePolicy(qf=qf, start_epsilon=0.25, end_epsilon=0.25, decay_frames=10, action_shape=A)

    class Stub:
        epoch_frames = 0

    class Log:
        def add_update_info(self, d): pass
        def add_epoch_info(self, *a, **k): pass
        def log(self, *a): pass
        def finish(self): pass
    kw = dict(qf=qf, pf=pf, qlr=2.5e-4, env=env, replay_buffer=None, collector=Stub(), logger=Log(), discount=0.99,
              num_epochs=10, batch_size=B, device=DEV, save_dir=None, tau=0.005, use_soft_update=True, opt_times=1)
    agent = QRDQN(quantile_num=Q, **kw) if Q > 1 else DQN(**kw)
    for s, batch in enumerate(dqn_batches(g, tag)):
        info = agent.update({k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()})
        ref = dict(zip([str(k) for k in g[f"{tag}_s{s}_info_keys"]], g[f"{tag}_s{s}_info_vals"]))
        assert sorted(info.keys()) == sorted(ref.keys())
        for k in ("Reward_Mean", "Training/qf_loss", "q_s_a", "epsilon"):
            assert abs(info[k] - ref[k]) < 2e-4 * abs(ref[k]) + 2e-6, (k, info[k], ref[k])
    for name, mod in (("qf1", qf), ("tqf1", agent.target_qf)):
        for k, p in mod.state_dict().items():
            err = np.abs(p.cpu().numpy() - g[f"{tag}_{name}_{k.replace('.', '__')}"]).max()
            assert err < 3e-6, (name, k, err)


def test_frame_env_policy_and_collector_vs_oracle():
    from torchrl.collector import VecCollector
    from torchrl.env import get_vec_env
    from torchrl.policies import EpsilonGreedyDQNDiscretePolicy, EpsilonGreedyQRDQNDiscretePolicy
    from torchrl.replay_buffers import BaseReplayBuffer
    from torchrl_amd import ops
    N, A, horizon, seed = 6, 6, 5, 2
    env = get_vec_env("SynthAtari-v0", {"reward_scale": 1}, N)
    eval_env = get_vec_env("SynthAtari-v0", {"reward_scale": 1}, N)
    env.horizon = eval_env.horizon = horizon
    env.seed(seed)
    oenv = SynthFrameVecEnvCPU(N, horizon=horizon)
    oenv.seed(seed)
    assert np.array_equal(env.reset().cpu().numpy(), oenv.reset())
    torch.manual_seed(3)
    qf = small_qnet(A).to(DEV)
    params = [p.detach().cpu() for p in ops.cnn_param_list(qf)]
    pf = EpsilonGreedyDQNDiscretePolicy(qf=qf, start_epsilon=0.5, end_epsilon=0.1, decay_frames=100, action_shape=A)
    buf = BaseReplayBuffer(N * 4, env_nums=N)                              # 4-row ring, 7 steps wrap
    col = VecCollector(env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=DEV, epoch_frames=N * 7,
                       max_episode_frames=4, eval_episodes=1)
    np.random.seed(11)
    got = col.train_one_epoch()
    # oracle replay of the same loop (reference: discrete_policies.py:43-67, collector/base.py:184-230)
    np.random.seed(11)
    obs = oenv.reset()
    cur = np.zeros(N)
    rows, total, count = [], 0.0, 0
    for t in range(7):
        count += 1
        eps = 0.5 - 0.4 * count / 100
        with torch.no_grad():
            q = cnn(scale_frames(obs), params, [4, 2, 1], "tanh")
        act = q.max(dim=-1)[1].numpy()
        r = np.random.rand(N, 1)
        ra = np.random.randint(0, A, size=(N, 1))
        act = np.where(r[:, 0] < eps, ra[:, 0], act)
        nobs, rew, done, _ = oenv.step(act)
        cur += 1
        rows.append((obs.copy(), nobs.copy(), act.copy(), rew.copy(), done.copy()))
        total += rew.sum()
        flag = done[:, 0] | (cur >= 4)
        if flag.any():
            nobs = oenv.reset(mask=flag)
            cur[flag] = 0
        obs = nobs
    assert abs(got["train_epoch_reward"] - total) < 1e-6
    for t in range(3, 7):                                                   # last 4 steps live in the ring
        row = t % 4
        o, no, a, rw, dn = rows[t]
        assert np.array_equal(buf._obs[row].cpu().numpy(), o) and np.array_equal(buf._next_obs[row].cpu().numpy(), no)
        assert np.array_equal(buf._acts[row].cpu().numpy()[:, 0], a.astype(np.float32))
        assert np.array_equal(buf._rewards[row].cpu().numpy(), rw) and np.array_equal(buf._terminals[row].cpu().numpy(), dn.astype(np.float32))
    assert buf._obs.dtype == torch.uint8 and np.array_equal(env.cur_obs.cpu().numpy(), obs)
    # QR-DQN greedy action = argmax of the quantile mean, vectorised over envs
    qfq = small_qnet(A, 8).to(DEV)
    pq = EpsilonGreedyQRDQNDiscretePolicy(quantile_num=8, qf=qfq, start_epsilon=0.0, end_epsilon=0.0, decay_frames=1, action_shape=A)
    qv = ops.cnn_forward(qfq, env.cur_obs)[0]
    want = qv.view(N, A, 8).mean(-1).max(-1)[1].cpu().numpy()
    assert np.array_equal(pq.eval_act(env.cur_obs)[:, 0], want)
    ev = col.eval_one_epoch()
    assert len(ev["eval_rewards"]) == N and ev["eval_traj_length"] == horizon


@pytest.mark.parametrize("Cout,group", [(64, False), (32, False), (64, True)])
def test_last_conv_layer_stores_the_flattened_features_itself(Cout, group):
    """`out_chw` of the channels-last conv layer: the same values as the NHWC result, stored in nn.Flatten's (c, oy, ox)
    order by the layer's epilogue (bit for bit what the transposing launch produced), at sizes with a ragged last tile;
    and the un-flattening transpose of the backward pass gated by an output kept in that order."""
    from torchrl_amd import _C
    torch.manual_seed(5)
    B, H, W, Cin, kh, kw, sh, sw = 37, 9, 9, 32, 3, 3, 1, 1
    x = torch.randn(B, H, W, Cin, device=DEV)
    ws = [torch.randn(Cout, Cin * kh * kw, device=DEV) * 0.05 for _ in range(2)]
    bs = [torch.randn(Cout, device=DEV) for _ in range(2)]
    if group:
        xs = [x, torch.randn_like(x)]
        want, (_, Ho, Wo) = _C.conv_fwd_nhwc_group(xs, ws, bs, kh, kw, sh, sw, _C.ACT_RELU)
        got, _ = _C.conv_fwd_nhwc_group(xs, ws, bs, kh, kw, sh, sw, _C.ACT_RELU, out_chw=True)
    else:
        w0, (_, Ho, Wo) = _C.conv_fwd_nhwc(x, ws[0], bs[0], kh, kw, sh, sw, _C.ACT_RELU)
        g0, _ = _C.conv_fwd_nhwc(x, ws[0], bs[0], kh, kw, sh, sw, _C.ACT_RELU, out_chw=True)
        want, got = [w0], [g0]
    P = Ho * Wo
    for w, g in zip(want, got):
        assert g.shape == (B, Cout * P)
        assert torch.equal(g.view(B, Cout, P), w.view(B, P, Cout).transpose(1, 2))
    d = torch.randn(B, Cout * P, device=DEV)
    a = _C.transpose_bpc(d.view(B, Cout, P), B, Cout, P, y_gate=want[0], gate_act=_C.ACT_RELU)
    b = _C.transpose_bpc(d.view(B, Cout, P), B, Cout, P, y_gate=got[0], gate_act=_C.ACT_RELU, gate_like_in=True)
    assert torch.equal(a, b)


@pytest.mark.parametrize("B,H,A,float_acts", [(512, 512, 6, True), (37, 64, 3, False), (130, 1024, 8, True), (1100, 256, 2, False), (5, 4, 1, False)])
def test_one_launch_dqn_head_equals_the_layer_kernels_and_the_loss_launch(B, H, A, float_acts):
    """`trl_dqn_head_f32` (both head forwards, the TD loss and the head's backward in one launch) against the launches
    it replaces -- linear_fwd x 2, dqn_td_loss, linear_bwd_weight, linear_bwd_input -- and against autograd in float64;
    run twice on one workspace (the arrival counter carries over) with bit-identical results."""
    from torchrl_amd import _C
    torch.manual_seed(B + A)
    h, hn = torch.randn(B, H, device=DEV), torch.randn(B, H, device=DEV)
    w, wt = torch.randn(A, H, device=DEV) * 0.1, torch.randn(A, H, device=DEV) * 0.1
    b, bt = torch.randn(A, device=DEV), torch.randn(A, device=DEV)
    acts_i = torch.randint(0, A, (B,), device=DEV)
    acts = acts_i.float() if float_acts else acts_i
    rew, term = torch.randn(B, device=DEV), (torch.rand(B, device=DEV) < 0.2).float()
    gamma = 0.97
    # the separate launches
    q, qn = _C.linear_fwd(h, w, b, _C.ACT_NONE), _C.linear_fwd(hn, wt, bt, _C.ACT_NONE)
    sums0 = torch.zeros(3, dtype=torch.float64, device=DEV)
    dq = _C.dqn_td_loss(q, acts, qn, rew, term, gamma, sums0)
    dw0, db0 = torch.zeros_like(w), torch.zeros_like(b)
    _C.linear_bwd_weight(dq, None, _C.ACT_NONE, h, dw=dw0, db=db0)
    dh0 = _C.linear_bwd_input(dq, None, _C.ACT_NONE, w)
    # one launch
    ws = _C.dqn_head_workspace(H, A, DEV)
    outs = []
    for _ in range(2):
        sums, dw, db = torch.zeros(3, dtype=torch.float64, device=DEV), torch.zeros_like(w), torch.zeros_like(b)
        dh, q1, qn1 = _C.dqn_head(h, hn, w, b, wt, bt, acts, rew, term, gamma, dw, db, sums, ws, want_q=True)
        outs.append((dh, q1, qn1, dw, db, sums))
    for x, y in zip(outs[0], outs[1]):
        assert torch.equal(x, y)
    dh, q1, qn1, dw, db, sums = outs[0]
    torch.testing.assert_close(q1, q, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(qn1, qn, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dh, dh0, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(dw, dw0, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(db, db0, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(sums, sums0, rtol=1e-5, atol=1e-7)
    # float64 autograd
    h64, w64, b64 = h.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    qq = h64 @ w64.t() + b64
    tgt = rew.double() + gamma * (1 - term.double()) * (hn.double() @ wt.double().t() + bt.double()).max(1).values
    loss = ((qq.gather(1, acts_i[:, None]).squeeze(1) - tgt) ** 2).mean()
    loss.backward()
    torch.testing.assert_close(dh.double(), h64.grad, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(dw.double(), w64.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(db.double(), b64.grad, rtol=1e-4, atol=1e-6)
    assert abs(sums[0].item() / B - loss.item()) <= 1e-5 * max(1.0, abs(loss.item()))


@pytest.mark.parametrize("Q,horizon,max_frames", [(1, 3, 5), (8, 5, 4)])      # episodes end by `done` / by the frame cap
def test_frame_env_rollout_as_a_replayed_graph_equals_the_step_by_step_rollout(Q, horizon, max_frames, monkeypatch):
    """collector/base.py _replayed_rollout_frames: host draws of all steps staged ahead, the steps of an epoch as ONE graph
    whose frame step files each transition into the ring row a device counter points at (trl_synth_frames_collect_u8; the
    ring wraps inside an epoch here).  Every epoch (eager first visit, captured, replayed ...) must leave the ring, the
    epoch results, the env and the policy's epsilon exactly as the step-by-step path does."""
    from torchrl.collector import VecCollector
    from torchrl.env import get_vec_env
    from torchrl.policies import EpsilonGreedyDQNDiscretePolicy, EpsilonGreedyQRDQNDiscretePolicy
    from torchrl.replay_buffers import BaseReplayBuffer
    N, A, steps, rows, epochs = 6, 5, 3, 7, 6

    def run(no_graph):
        if no_graph:
            monkeypatch.setenv("TRL_NO_GRAPH", "1")
        else:
            monkeypatch.delenv("TRL_NO_GRAPH", raising=False)
        env = get_vec_env("SynthAtari-v0", {"reward_scale": 1}, N)
        eval_env = get_vec_env("SynthAtari-v0", {"reward_scale": 1}, N)
        env.horizon = eval_env.horizon = horizon
        env.seed(4)
        env.reset()
        torch.manual_seed(5)
        qf = small_qnet(A, Q).to(DEV)
        kw = dict(qf=qf, start_epsilon=0.6, end_epsilon=0.2, decay_frames=10, action_shape=A)
        pf = EpsilonGreedyQRDQNDiscretePolicy(quantile_num=Q, **kw) if Q > 1 else EpsilonGreedyDQNDiscretePolicy(**kw)
        buf = BaseReplayBuffer(N * rows, env_nums=N)
        col = VecCollector(env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=DEV, epoch_frames=N * steps,
                           max_episode_frames=max_frames, eval_episodes=1)
        np.random.seed(12)
        out = []
        for e in range(epochs):
            res = col.train_one_epoch()
            out.append((float(res["train_epoch_reward"]), [float(x) for x in res["train_rewards"]], pf.epsilon, pf.count,
                        buf._top, buf._size, col.global_step,
                        {k: getattr(buf, "_" + k).cpu().numpy().copy() for k in ("obs", "next_obs", "acts", "rewards", "terminals", "time_limits")},
                        env.cur_obs.cpu().numpy().copy(), env.t_env.cpu().numpy().copy(), env.cur_step.cpu().numpy().copy(),
                        env.ep_return.cpu().numpy().copy()))
        ev = col.eval_one_epoch()
        return out, getattr(col, "_fr", None), ev, np.random.rand(3)

    plain, fr0, ev0, tail0 = run(True)
    graph, fr1, ev1, tail1 = run(False)
    assert fr0 is None and fr1 is not None and fr1["graph"] is not None        # the second run did replay a graph
    assert np.array_equal(tail0, tail1)                                          # the numpy stream is where it would be
    assert ev0["eval_rewards"] == ev1["eval_rewards"]
    for e in range(epochs):
        a, b = plain[e], graph[e]
        assert a[:7] == b[:7], (e, a[:7], b[:7])
        for k in a[7]:
            assert np.array_equal(a[7][k], b[7][k]), (e, k)
        for i in range(8, 12):
            assert np.array_equal(a[i], b[i]), (e, i)
    assert any(len(p[1]) for p in plain) == (horizon < max_frames)               # finished episodes were logged along the way


@pytest.mark.parametrize("B", [5, 64])
def test_conv_weight_gradient_folds_as_one_launch_equal_the_per_layer_folds(B):
    """_C.FoldScope: the folds of the conv trunk's split weight-gradient partials recorded and run as ONE launch
    (trl_fold_scope_*) against the per-layer launches -- same arithmetic, same order: bit-identical gradients.  The scope
    closes on an error inside, and does not nest."""
    from torchrl_amd import _C, ops
    torch.manual_seed(2)
    net = small_qnet(6, act=torch.nn.ReLU).to(DEV)
    frames = torch.randint(0, 256, (B, 4, 84, 84), dtype=torch.uint8, device=DEV)
    plist = ops.cnn_param_list(net)
    out, tape = ops.cnn_forward(net, frames)
    d_out = torch.randn(B, 6, device=DEV)
    mk = lambda: [(torch.zeros_like(plist[k]), torch.zeros_like(plist[k + 1])) for k in range(0, len(plist), 2)]
    one, per_layer = mk(), mk()
    ops.cnn_backward(net, tape, d_out, one)                               # folds together at the end of the trunk
    # the same trunk without the scope: every layer folds in its own launch
    n_conv = len(tape.convs)
    d_feat = ops.mlp_backward(tape.fc, d_out, grads=per_layer[n_conv:], need_input=True)
    P, Cc = tape.feat_shape
    d = _C.transpose_bpc(d_feat.view(tape.B, Cc, P), tape.B, Cc, P, y_gate=tape.convs[-1][2], gate_act=tape.act,
                         gate_like_in=tape.feat_chw).view(tape.B * P, Cc)
    needs = ops._conv_ws_needs(tape)
    ops._cnn_trunk_backward(tape, d, per_layer, True, [torch.empty(n, device=DEV) for n in needs])
    for (w1, b1), (w2, b2) in zip(one, per_layer):
        assert torch.equal(w1, w2) and torch.equal(b1, b2)
    assert any(w.abs().sum().item() > 0 for w, _ in one[:n_conv])
    with pytest.raises(RuntimeError):
        with _C.FoldScope(DEV):
            raise RuntimeError("inside")
    with _C.FoldScope(DEV):                                                # (closed by the failure above: opens again)
        with pytest.raises(_C.TrlError):
            _C.check(_C.lib().trl_fold_scope_begin(), "nested")


@pytest.mark.parametrize("B", [3, 64])
def test_forward_with_reordered_weight_shadows_equals_the_gathering_kernel(B, monkeypatch):
    """ops.FWD_WEIGHT_SHADOWS: the later conv layers' weights re-ordered to the (i, j, c) reduction order by riders of the
    first layer's launch (trl_conv_fwd_u8_f32 perm jobs, trl_conv_fwd_nhwc_f32 w_perm) -- same products in the same order as
    the kernel that gathers nn.Conv2d's layout on the fly: bit-identical outputs and tapes, single and paired forward; the
    shadows follow the live weights (changed in between)."""
    from torchrl_amd import _C, ops
    torch.manual_seed(B)
    net, net_t = small_qnet(6, act=torch.nn.ReLU).to(DEV), small_qnet(6, act=torch.nn.ReLU).to(DEV)
    fa = torch.randint(0, 256, (B, 4, 84, 84), dtype=torch.uint8, device=DEV)
    fb = torch.randint(0, 256, (B, 4, 84, 84), dtype=torch.uint8, device=DEV)

    def run():
        out, tape = ops.cnn_forward(net, fa)
        (oa, ta), (ob, tb) = ops.cnn_forward_pair(net, net_t, fa, fb)
        return [out, oa, ob] + [c[2] for c in tape.convs] + [c[2] for c in ta.convs] + [c[2] for c in tb.convs]

    for step in range(2):
        monkeypatch.setattr(ops, "FWD_WEIGHT_SHADOWS", False)
        want = run()
        monkeypatch.setattr(ops, "FWD_WEIGHT_SHADOWS", True)
        got = run()
        assert len(want) == len(got) and all(torch.equal(w, g) for w, g in zip(want, got))
        with torch.no_grad():                                               # new weights: the next pass re-makes its shadows
            for p in list(net.parameters()) + list(net_t.parameters()):
                p.add_(0.01 * torch.randn_like(p))
    # the re-ordering itself, against torch
    w = torch.randn(32, 16 * 3 * 3, device=DEV)
    frames = torch.zeros(2, 4, 84, 84, dtype=torch.uint8, device=DEV)
    w1 = torch.zeros(8, 4 * 8 * 8, device=DEV)
    w2 = torch.randn(32, 16 * 4 * 4, device=DEV)
    _, _, (outs, wss) = _C.conv_fwd_u8(frames, w1, None, 8, 8, 4, 4, 1.0, 0.0, _C.ACT_NONE, perm=[(w, 16, 9)],
                                       dx=[(w2, 16, 4, 4, 2, 2), (w, 16, 3, 3, 1, 1)])
    assert torch.equal(outs[0], w.view(32, 16, 9).permute(0, 2, 1).reshape(32, -1))
    # ... and the input-gradient kernels' weight re-orderings as riders of the same launch, against the launch of their own
    ref = _C.conv_bwd_input_prep([(w2, 16, 4, 4, 2, 2), (w, 16, 3, 3, 1, 1)], DEV)
    assert torch.equal(wss[0], ref[0]) and torch.equal(wss[1], ref[1])


@pytest.mark.parametrize("M,K,N,act", [(512, 512, 1200, "relu"), (100, 70, 1030, "tanh"), (64, 64, 2048, "none")])
def test_input_gradient_with_the_reduction_split_over_slices(M, K, N, act):
    """trl_linear_bwd_input_splitk_f32 (few output tiles behind a long reduction: a wide head's input gradient) against
    float64 and against the unsplit launch; deterministic."""
    from torchrl_amd import _C
    gen = torch.Generator().manual_seed(M + N)
    dy, y, w = (torch.randn(M, N, generator=gen).to(DEV), torch.tanh(torch.randn(M, N, generator=gen)).to(DEV),
                (torch.randn(N, K, generator=gen) / N ** 0.5).to(DEV))
    if act == "relu":
        y = y.clamp_min(0.0)
    code = {"relu": _C.ACT_RELU, "tanh": _C.ACT_TANH, "none": _C.ACT_NONE}[act]
    gate = None if act == "none" else y
    assert _C.lib().trl_linear_bwd_input_workspace(M, K, N) > 0
    got = _C.linear_bwd_input(dy, gate, code, w)                          # takes the split path
    again = _C.linear_bwd_input(dy, gate, code, w)
    assert torch.equal(got, again)
    plain = torch.empty(M, K, device=DEV)
    _C.check(_C.lib().trl_linear_bwd_input_f32(_C.dev_ptr(dy), _C.dev_ptr(gate, allow_none=True), code, _C.dev_ptr(w),
                                               _C.dev_ptr(plain), M, K, N, _C.stream_ptr(DEV)), "plain")
    d = {"relu": (y > 0).double(), "tanh": 1.0 - y.double() ** 2, "none": torch.ones_like(y).double()}[act]
    want = (dy.double() * d) @ w.double()
    sc = max(1.0, want.abs().max().item())
    assert (got.double() - want).abs().max().item() < 2e-5 * sc and (plain.double() - want).abs().max().item() < 2e-5 * sc
    assert _C.lib().trl_linear_bwd_input_workspace(4096, 256, 256) == 0    # ordinary layers do not split


@pytest.mark.parametrize("N,H,A", [(512, 512, 6), (7, 32, 3), (130, 1024, 8), (3, 4, 1)])
def test_head_and_action_in_one_launch_equal_the_layer_and_the_action_kernel(N, H, A):
    """trl_dqn_act_f32: q = h W^T + b within round-off of the dense layer (another summation order), the action exactly what
    trl_eps_greedy_i64 makes of THESE q values (greedy and mixed), deterministic."""
    from torchrl_amd import _C
    gen = torch.Generator().manual_seed(N + H)
    h, w, b = torch.randn(N, H, generator=gen).to(DEV), (torch.randn(A, H, generator=gen) / H ** 0.5).to(DEV), torch.randn(A, generator=gen).to(DEV)
    u, ra = torch.rand(N, generator=gen).to(DEV), torch.randint(0, A, (N,), generator=gen).to(DEV)
    assert _C.dqn_act_ok(h, w)
    q, act = _C.dqn_act(h, w, b, u, ra, 0.3)
    q2, act2 = _C.dqn_act(h, w, b, u, ra, 0.3)
    assert torch.equal(q, q2) and torch.equal(act, act2)
    ref = _C.linear_fwd(h, w, b, _C.ACT_NONE)
    assert (q - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    assert torch.equal(act, _C.eps_greedy(q, A, 1, u, ra, 0.3))
    _, greedy = _C.dqn_act(h, w, b, None, None, 0.0, want_q=False)
    assert torch.equal(greedy, q.max(dim=-1)[1])
