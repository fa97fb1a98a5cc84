"""Twin-Q SAC path on the GPU: generic MFMA dense-layer kernels vs torch, the SAC element-wise
kernels and the whole TwinSACQ.update vs the reference's outputs (tests/golden/twin_sac_q.npz),
the off-policy collector vs the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import nets, replay
from oracle.collector import VecCollectorOracle
from oracle.sac import rsample
from oracle.synth_env import SynthVecEnvCPU

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("M,K,N", [(256, 23, 256), (4096, 256, 256), (100, 17, 12), (65, 256, 1), (1, 5, 3),
                                   (300, 132, 70), (515, 392, 96), (130, 260, 68), (200, 3136, 96), (64, 1100, 130)])
@pytest.mark.parametrize("act", ["tanh", "relu", "none"])
def test_linear_layer_kernels_vs_torch(M, K, N, act):
    from torchrl_amd import _C
    code = {"tanh": _C.ACT_TANH, "relu": _C.ACT_RELU, "none": _C.ACT_NONE}[act]
    gen = torch.Generator().manual_seed(M + K + N)
    x, w, b = torch.randn(M, K, generator=gen), torch.randn(N, K, generator=gen) / K ** 0.5, torch.randn(N, generator=gen)
    dy = torch.randn(M, N, generator=gen)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    z = torch.nn.functional.linear(xr, wr, br)
    y = {"tanh": torch.tanh, "relu": torch.relu, "none": lambda t: t}[act](z)
    y.backward(dy)
    yd = _C.linear_fwd(x.to(DEV), w.to(DEV), b.to(DEV), code)
    scale = lambda t: max(1.0, t.abs().max().item())
    assert (yd.cpu() - y.detach()).abs().max().item() < 2e-5 * scale(y.detach())
    gate = None if act == "none" else yd
    dx = _C.linear_bwd_input(dy.to(DEV), gate, code, w.to(DEV))
    assert (dx.cpu() - xr.grad).abs().max().item() < 5e-5 * scale(xr.grad)
    dw, db = _C.linear_bwd_weight(dy.to(DEV), gate, code, x.to(DEV))
    assert (dw.cpu() - wr.grad).abs().max().item() < 5e-5 * scale(wr.grad)
    assert (db.cpu() - br.grad).abs().max().item() < 5e-5 * scale(br.grad)
    # no bias / bias gradient not requested
    assert (_C.linear_fwd(x.to(DEV), w.to(DEV), None, _C.ACT_NONE).cpu() - x @ w.t()).abs().max().item() < 2e-5 * scale(y.detach())
    assert _C.linear_bwd_weight(dy.to(DEV), gate, code, x.to(DEV), need_bias=False)[1] is None


@pytest.mark.parametrize("M,K,N,G", [(4096, 256, 256, 6), (300, 23, 70, 2), (130, 256, 1, 3)])
def test_grouped_layer_launches_equal_separate_calls(M, K, N, G):
    """trl_linear_*_group_f32: G same-shaped layers in one launch give bit-identical results to G separate calls
    (same kernel, same summation order) -- shared inputs, distinct weights, gated and ungated."""
    from torchrl_amd import _C
    gen = torch.Generator().manual_seed(M + G)
    xs = [torch.randn(M, K, generator=gen).to(DEV) for _ in range(G)]
    xs[1] = xs[0]                                                    # a shared input (Q1 / Q2 on the same batch)
    ws = [(torch.randn(N, K, generator=gen) / K ** 0.5).to(DEV) for _ in range(G)]
    bs = [torch.randn(N, generator=gen).to(DEV) for _ in range(G)]
    dys = [torch.randn(M, N, generator=gen).to(DEV) for _ in range(G)]
    ys = _C.linear_fwd_group(xs, ws, bs, _C.ACT_RELU)
    for g in range(G):
        assert torch.equal(ys[g], _C.linear_fwd(xs[g], ws[g], bs[g], _C.ACT_RELU))
    assert torch.equal(_C.linear_fwd_group(xs, ws, [None] * G, _C.ACT_NONE)[G - 1], _C.linear_fwd(xs[-1], ws[-1], None, _C.ACT_NONE))
    dxs = _C.linear_bwd_input_group(dys, ys, _C.ACT_RELU, ws)
    dxs_plain = _C.linear_bwd_input_group(dys, [None] * G, _C.ACT_NONE, ws)
    dws = [torch.empty(N, K, device=DEV) for _ in range(G)]
    dbs = [torch.empty(N, device=DEV) for _ in range(G)]
    _C.linear_bwd_weight_group(dys, ys, _C.ACT_RELU, xs, dws, dbs)
    for g in range(G):
        assert torch.equal(dxs[g], _C.linear_bwd_input(dys[g], ys[g], _C.ACT_RELU, ws[g]))
        assert torch.equal(dxs_plain[g], _C.linear_bwd_input(dys[g], None, _C.ACT_NONE, ws[g]))
        dw, db = _C.linear_bwd_weight(dys[g], ys[g], _C.ACT_RELU, xs[g])
        assert torch.equal(dws[g], dw) and torch.equal(dbs[g], db)


@pytest.mark.parametrize("M,K,N,G", [(4096, 256, 256, 2), (4096, 23, 256, 2), (4096, 256, 1, 2), (300, 17, 12, 1)])
def test_deferred_folds_equal_the_folding_entry_point(M, K, N, G):
    """Split weight-gradient GEMMs whose partials are folded later, several layers per launch
    (trl_linear_bwd_weight_partials_group_f32 + trl_fold_partials_multi_f32), against the entry point that folds at
    once: the same partials summed in the same order."""
    from torchrl_amd import _C
    gen = torch.Generator().manual_seed(M + K + N)
    r = lambda *s: torch.randn(*s, generator=gen).to(DEV)
    dys, ys, xs = [r(M, N) for _ in range(G)], [r(M, N) for _ in range(G)], [r(M, K) for _ in range(G)]
    want_w, want_b = [torch.empty(N, K, device=DEV) for _ in range(G)], [torch.empty(N, device=DEV) for _ in range(G)]
    _C.linear_bwd_weight_group(dys, ys, _C.ACT_RELU, xs, want_w, want_b)
    need = 2 * G * _C.lib().trl_linear_bwd_weight_workspace(M, K, N)
    plan = _C.FoldPlan(torch.empty(need, device=DEV))
    got_w, got_b = [torch.zeros(N, K, device=DEV) for _ in range(G)], [torch.zeros(N, device=DEV) for _ in range(G)]
    other_w, other_b = [torch.zeros(N, K, device=DEV) for _ in range(G)], [torch.zeros(N, device=DEV) for _ in range(G)]
    _C.linear_bwd_weight_partials_group(dys, ys, _C.ACT_RELU, xs, got_w, got_b, plan)
    _C.linear_bwd_weight_partials_group(dys[::-1], ys[::-1], _C.ACT_RELU, xs[::-1], other_w, other_b, plan)   # a second "layer"
    assert len(plan.entries) == 4 * G
    plan.run()
    for g in range(G):
        assert torch.equal(got_w[g], want_w[g]) and torch.equal(got_b[g], want_b[g])
        assert torch.equal(other_w[g], want_w[G - 1 - g]) and torch.equal(other_b[g], want_b[G - 1 - g])
    with pytest.raises(_C.TrlError):
        _C.linear_bwd_weight_partials_group(dys, ys, _C.ACT_RELU, xs, got_w, got_b, _C.FoldPlan(torch.empty(8, device=DEV)))


def test_weight_gradients_of_layers_of_different_widths_in_one_launch():
    """trl_linear_bwd_weight_partials_multi_f32: nine layers of a SAC update (17 / 23 / 256 inputs, 256 / 12 / 1 outputs,
    gated and ungated) as one launch of split GEMMs + one fold, against one folding call per layer."""
    from torchrl_amd import _C
    M = 4096
    gen = torch.Generator().manual_seed(77)
    r = lambda *s: torch.randn(*s, generator=gen).to(DEV)
    shapes = [(17, 256, True), (256, 256, True), (256, 12, False), (23, 256, True), (256, 256, True), (256, 1, False),
              (23, 256, True), (256, 256, True), (256, 1, False), (70, 33, True)]
    need = sum(_C.lib().trl_linear_bwd_weight_multi_splits(M, K, N) * (N * K + N) for K, N, _ in shapes)
    plan = _C.FoldPlan(torch.empty(need, device=DEV), defer_gemm=True)
    want, got = [], []
    for K, N, gated in shapes:
        dy, y, x = r(M, N), r(M, N), r(M, K)
        ww, wb = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
        act = _C.ACT_RELU if gated else _C.ACT_NONE
        _C.linear_bwd_weight(dy, y if gated else None, act, x, dw=ww, db=wb)
        gw, gb = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
        _C.linear_bwd_weight_partials_group([dy], [y if gated else None], act, [x], [gw], [gb], plan)
        want.append((ww, wb)); got.append((gw, gb))
    assert len(plan.problems) == 10 and len(plan.entries) == 20
    plan.run()
    assert not plan.problems and not plan.entries
    for (ww, wb), (gw, gb), (K, N, _) in zip(want, got, shapes):
        scale = max(1.0, ww.abs().max().item())
        assert (gw - ww).abs().max().item() < 2e-5 * scale, (K, N)        # other split counts: another summation order
        assert (gb - wb).abs().max().item() < 2e-5 * max(1.0, wb.abs().max().item()), (K, N)


@pytest.mark.parametrize("M,D,O,act,last", [(4096, 23, 1, "relu", "none"), (4096, 17, 12, "relu", "none"),
                                            (1000, 17, 6, "tanh", "tanh"), (33, 32, 16, "relu", "none"), (1, 3, 1, "tanh", "none")])
def test_fused_three_layer_forward_equals_the_per_layer_launches(M, D, O, act, last, monkeypatch):
    """trl_mlp3_forward_group_f32 (D -> 256 -> 256 -> O, hidden activations on chip, several networks per launch, hidden
    tapes only where asked for) against three dense-layer launches: hidden activations bit for bit (same k order), the
    head within round-off (its reduction is split over four waves); networks at unaligned offsets of a flat block."""
    from torchrl_amd import _C, ops
    code = {"relu": _C.ACT_RELU, "tanh": _C.ACT_TANH, "none": _C.ACT_NONE}
    gen = torch.Generator().manual_seed(M + D + O)
    G = 3
    sizes = [256 * D, 256, 256 * 256, 256, O * 256, O]
    flat = torch.randn(G * sum(sizes) + 7, generator=gen).to(DEV) * 0.1
    layers_list, off = [], 1                                                # odd offset: the second network's W2 is unaligned
    for g in range(G):
        ps = []
        for n in sizes:
            ps.append(flat[off:off + n]); off += n
        layers_list.append([(ps[0].view(256, D), ps[1]), (ps[2].view(256, 256), ps[3]), (ps[4].view(O, 256), ps[5])])
        off += 1
    xs = [torch.randn(M, D, generator=gen).to(DEV) for _ in range(G)]
    assert _C.mlp3_forward_ok(D, 256, 256, O)
    keep = [True, False, True]
    outs, tapes = ops.mlp_forward_group(layers_list, xs, code[act], last_act=code[last], keep=keep)
    want, wtapes = [], []                                                  # the three dense-layer launches of each network
    for ls, x in zip(layers_list, xs):
        t = ops.Tape()
        t.x, t.layers, t.act, t.last_act, t.outs = x, ls, code[act], code[last], []
        h = x
        for k, (w, b) in enumerate(ls):
            h = _C.linear_fwd(h, w, b, code[last] if k == 2 else code[act])
            t.outs.append(h)
        want.append(h)
        wtapes.append(t)
    for g in range(G):
        if keep[g]:
            assert torch.equal(tapes[g].outs[0], wtapes[g].outs[0]) and torch.equal(tapes[g].outs[1], wtapes[g].outs[1])
        else:
            assert tapes[g].outs[0] is None and tapes[g].outs[1] is None
        err = (outs[g] - want[g]).abs().max().item()
        assert err < 2e-6 * max(1.0, want[g].abs().max().item()), (g, err)
    # the backward pass runs on a fused tape as on a per-layer one
    d = torch.randn(M, O, generator=gen).to(DEV)
    mk = lambda: [(torch.zeros(256, D, device=DEV), torch.zeros(256, device=DEV)),
                  (torch.zeros(256, 256, device=DEV), torch.zeros(256, device=DEV)),
                  (torch.zeros(O, 256, device=DEV), torch.zeros(O, device=DEV))]
    ga, gb = mk(), mk()
    if last == "none":
        dxa = ops.mlp_backward(tapes[0], d, grads=ga, need_input=True)
        dxb = ops.mlp_backward(wtapes[0], d, grads=gb, need_input=True)
        assert torch.equal(dxa, dxb)
        for (wa, ba), (wb, bb) in zip(ga, gb):
            assert torch.equal(wa, wb) and torch.equal(ba, bb)
    assert not _C.mlp3_forward_ok(40, 256, 256, 1) and not _C.mlp3_forward_ok(17, 64, 64, 6) and not _C.mlp3_forward_ok(17, 256, 256, 20)


@pytest.mark.parametrize("act", ["relu", "tanh"])
def test_one_output_head_backward_through_the_rank_one_kernel(act, monkeypatch):
    """mlp_backward_group on networks with a ONE-output head: d(hidden) = dq w^T comes from trl_outer_gate_group_f32 already
    gated, the layer below then runs ungated -- against float64 autograd: input gradients and every
    weight / bias gradient (twin group, one network at an odd parameter offset)."""
    from torchrl_amd import _C, ops
    code = {"relu": _C.ACT_RELU, "tanh": _C.ACT_TANH}[act]
    gen = torch.Generator().manual_seed(5)
    M, D, H, G = 520, 23, 256, 2
    sizes = [H * D, H, H * H, H, H, 1]
    flat = torch.randn(G * (sum(sizes) + 1) + 8, generator=gen).to(DEV) * 0.1
    layers_list, off = [], 0
    for g in range(G):
        ps = []
        for n in sizes:
            ps.append(flat[off:off + n]); off += n
        layers_list.append([(ps[0].view(H, D), ps[1]), (ps[2].view(H, H), ps[3]), (ps[4].view(1, H), ps[5])])
        off += 1                                                            # the second network starts at an odd offset
    xs = [torch.randn(M, D, generator=gen).to(DEV) for _ in range(G)]
    dqs = [torch.randn(M, 1, generator=gen).to(DEV) for _ in range(G)]
    outs, tapes = ops.mlp_forward_group(layers_list, xs, code)
    grads = [[(torch.zeros_like(w), torch.zeros_like(b)) for w, b in ls] for ls in layers_list]
    dxs = ops.mlp_backward_group(tapes, dqs, grads_list=grads, need_input=True)
    fn = torch.relu if act == "relu" else torch.tanh
    for g in range(G):                                                      # float64 autograd on the CPU as the reference
        ps = [p.detach().double().cpu().requires_grad_(True) for wb in layers_list[g] for p in wb]
        x = xs[g].double().cpu().requires_grad_(True)
        h = fn(fn(x @ ps[0].T + ps[1]) @ ps[2].T + ps[3])
        ((h @ ps[4].T + ps[5]) * dqs[g].double().cpu()).sum().backward()
        assert (dxs[g].double().cpu() - x.grad).abs().max().item() < 1e-5 * max(1.0, x.grad.abs().max().item())
        for (wa, ba), pw, pb in zip(grads[g], ps[0::2], ps[1::2]):
            assert (wa.double().cpu() - pw.grad).abs().max().item() < 2e-5 * max(1.0, pw.grad.abs().max().item())
            assert (ba.double().cpu() - pb.grad).abs().max().item() < 2e-5 * max(1.0, pb.grad.abs().max().item())


def test_rsample_fwd_bwd_vs_autograd():
    from torchrl_amd import _C
    B, A = 300, 6
    gen = torch.Generator().manual_seed(4)
    head = torch.randn(B, 2 * A, generator=gen)
    head[:, A:] = head[:, A:] * 0.5 - 1.5                          # moderate std: tanh rarely saturates
    head[0, A] = 3.0; head[1, A + 1] = -25.0                      # clamp boundaries: gradient must stop there
    eps, d_act = torch.randn(B, A, generator=gen), torch.randn(B, A, generator=gen)
    alpha, w_std, w_mean = 0.37, 1e-3, 2e-3
    hr = head.clone().requires_grad_(True)
    a, lp, mean, log_std = rsample(hr, eps)
    loss = (a * d_act).sum() + (alpha / B) * lp.sum() + w_std * (log_std ** 2).mean() + w_mean * (mean ** 2).mean()
    loss.backward()
    act, logp = _C.rsample_fwd(head.to(DEV), eps.to(DEV))
    np.testing.assert_allclose(act.cpu().numpy(), a.detach().numpy(), atol=1e-6)
    # log(1 - a^2 + 1e-6) is ill-conditioned once |a| -> 1 (the reference calls the formula "not very
    # numerically stable", distribution.py:11): compare rows whose actions are not saturated
    ok = (a.detach().abs().max(dim=1).values < 0.999).numpy()
    assert ok.mean() > 0.8
    np.testing.assert_allclose(logp.cpu().numpy()[ok], lp.detach().numpy()[ok, 0], rtol=2e-5, atol=2e-4)
    d_head = _C.rsample_bwd(head.to(DEV), eps.to(DEV), act, d_act.to(DEV), torch.tensor([alpha], device=DEV), 1.0 / B,
                            w_std, w_mean)
    okr = torch.as_tensor(ok)
    err = (d_head.cpu() - hr.grad)[okr].abs().max().item()
    assert err < 5e-5 * max(1.0, hr.grad[okr].abs().max().item()), err
    assert d_head[0, A].item() == 0.0 and d_head[1, A + 1].item() == 0.0


def test_one_launch_forms_equal_the_separate_kernels():
    """trl_sac_samples_f32 = 2 x rsample_fwd + 3 x concat2; trl_tanh_gauss_rsample_bwd_cols_f32 = slice_add +
    rsample_bwd; trl_moments_multi_f64 = 3 x moments: bit for bit (same arithmetic, fewer launches)."""
    from torchrl_amd import _C
    B, A, D = 777, 6, 17
    gen = torch.Generator().manual_seed(8)
    r = lambda *s: torch.randn(*s, generator=gen).to(DEV)
    head, head2, eps1, eps2, obs, acts, nobs = r(B, 2 * A), r(B, 2 * A), r(B, A), r(B, A), r(B, D), r(B, A), r(B, D)
    new_a, logp, next_a, next_logp, x_sa, x_next, x_new = _C.sac_samples(head, head2, eps1, eps2, obs, acts, nobs)
    a1, l1 = _C.rsample_fwd(head, eps1)
    a2, l2 = _C.rsample_fwd(head2, eps2)
    assert torch.equal(new_a, a1) and torch.equal(logp, l1) and torch.equal(next_a, a2) and torch.equal(next_logp, l2)
    assert torch.equal(x_sa, _C.concat2(obs, acts)) and torch.equal(x_next, _C.concat2(nobs, a2))
    assert torch.equal(x_new, _C.concat2(obs, a1))
    dx1, dx2, alpha = r(B, D + A), r(B, D + A), torch.tensor([0.4], device=DEV)
    want = _C.rsample_bwd(head, eps1, a1, _C.slice_add(dx1, dx2, D, A), alpha, 1.0 / B, 1e-3, 2e-3)
    assert torch.equal(_C.rsample_bwd_cols(head, eps1, a1, dx1, dx2, D, alpha, 1.0 / B, 1e-3, 2e-3), want)
    one = _C.rsample_bwd(head, eps1, a1, _C.slice_add(dx1, None, D, A), alpha, 1.0 / B, 0.0, 0.0)
    assert torch.equal(_C.rsample_bwd_cols(head, eps1, a1, dx1, None, D, alpha, 1.0 / B, 0.0, 0.0), one)
    got, ref = torch.zeros(3, 4, dtype=torch.float64, device=DEV), torch.zeros(3, 4, dtype=torch.float64, device=DEV)
    inf = float("inf")
    _C.moments_multi([(head, got[0], 2 * A, A, A, -20.0, 2.0), (logp, got[1], 1, 0, 1, -inf, inf),
                      (head, got[2], 2 * A, 0, A, -inf, inf)])
    _C.moments(head, ref[0], ld=2 * A, off=A, width=A, lo=-20.0, hi=2.0)
    _C.moments(logp, ref[1], ld=1)
    _C.moments(head, ref[2], ld=2 * A, off=0, width=A)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("n", [300, 40000, 400000])
def test_clip_adam_advances_the_device_step_itself(n):
    """Device-resident step state {t, beta1^t, beta2^t}: small grids advance it themselves (block count), large grids
    through the one-thread tick launch; three calls leave exactly three steps of torch.optim.Adam."""
    from torchrl_amd import _C
    gen = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=gen)
    gs = [torch.randn(n, generator=gen) for _ in range(3)]
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-2)
    p, g = p0.to(DEV).clone(), torch.zeros(n, device=DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    state = torch.tensor([0.0, 1.0, 1.0, 0.0], dtype=torch.float64, device=DEV)
    a = _C.AdamArgs()
    a.params, a.grads, a.exp_avg, a.exp_avg_sq = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
    a.n_groups = 1
    a.group_sizes[0] = n
    a.group_lr[0] = 1e-2
    a.max_norm, a.beta1, a.beta2, a.eps, a.grad_scale = 0.0, 0.9, 0.999, 1e-8, 1.0
    a.step_count, a.norms_out, a.step_state = 0, None, state.data_ptr()
    for k in range(3):
        g.copy_(gs[k])
        _C.clip_adam(a, torch.device(DEV))
        ref.grad = gs[k].clone()
        opt.step()
        st = state.cpu()
        assert st[0].item() == k + 1 and abs(st[1].item() - float(np.float32(0.9)) ** (k + 1)) < 1e-12
        assert state[3:].view(torch.int32)[0].item() == 0               # the block counter is back at zero
    assert (p.cpu() - ref.detach()).abs().max().item() < 2e-6


@pytest.mark.parametrize("sizes,clip", [((5000, 3000, 7000), 1.0), ((217000 // 3, 70000, 74000), 0.5), ((300, 200, 100), 0.0)])
def test_one_launch_update_tail_equals_fold_clip_adam_polyak_as_separate_launches(sizes, clip):
    """`trl_fold_clip_adam_polyak_f32` against trl_fold_partials_multi_f32 -> trl_clip_adam_polyak_f32 on the same
    partials: folded gradients bit for bit (same summation order), norms / parameters / moments / targets to round-off
    (the norm is summed in another order), the step state advanced once per call, the statistics block filed."""
    import ctypes as C
    from torchrl_amd import _C
    dev = torch.device(DEV)
    torch.manual_seed(sum(sizes))
    total = sum(sizes)
    # entries: every group is cut into a few "layers" with their own number of splits
    ents, pos = [], 0
    for gsz in sizes:
        left = gsz
        for frac, splits in ((0.6, 16), (0.3, 32), (0.1, 5)):
            n = max(1, int(gsz * frac)) if frac != 0.1 else left
            n = min(n, left)
            if n > 0:
                ents.append((pos, n, splits))
                pos += n; left -= n
    parts = [torch.randn(sp, n, device=dev) * 0.1 for _, n, sp in ents]
    p0 = torch.randn(total, device=dev)
    t_off, t_n = sizes[0], total - sizes[0]
    tgt0 = torch.randn(t_n, device=dev)
    raw = torch.arange(64, dtype=torch.uint8, device=dev)

    def make(p, g, m, v, state, norms):
        a = _C.AdamArgs()
        a.params, a.grads, a.exp_avg, a.exp_avg_sq = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
        a.n_groups = 3
        for k in range(3):
            a.group_sizes[k] = sizes[k]
            a.group_lr[k] = (3e-4, 1e-3, 2e-3)[k]
        a.max_norm, a.beta1, a.beta2, a.eps, a.grad_scale = clip, 0.9, 0.999, 1e-8, 1.0
        a.step_count, a.norms_out, a.step_state = 0, norms.data_ptr(), state.data_ptr()
        return a

    outs = []
    for fused in (False, True):
        p, g = p0.clone(), torch.zeros(total, device=dev)
        m, v, tgt = torch.zeros(total, device=dev), torch.zeros(total, device=dev), tgt0.clone()
        state = torch.tensor([0.0, 1.0, 1.0, 0.0], dtype=torch.float64, device=dev)
        norms, ring = torch.zeros(3, device=dev), torch.zeros(4, 64, dtype=torch.uint8, device=dev)
        a = make(p, g, m, v, state, norms)
        ws = _C.fold_clip_adam_polyak_workspace(dev)
        c = len(ents)
        for step in range(2):
            pp, nn, ss = (C.c_void_p * c)(), (C.c_int * c)(), (C.c_int * c)()
            oo = (C.c_void_p * c)()
            for j, ((off, n, sp), part) in enumerate(zip(ents, parts)):
                pp[j], nn[j], ss[j], oo[j] = part.data_ptr(), n, sp, g.data_ptr() + 4 * off
            if fused:
                _C.check(_C.lib().trl_fold_clip_adam_polyak_f32(c, pp, nn, ss, C.byref(a), tgt.data_ptr(), t_off, t_n, 0.005,
                                                                raw.data_ptr(), 64, ring.data_ptr(), 4, ws.data_ptr(),
                                                                _C.stream_ptr(dev)), "fused tail")
            else:
                _C.check(_C.lib().trl_fold_partials_multi_f32(c, pp, oo, nn, ss, _C.stream_ptr(dev)), "fold")
                _C.clip_adam_polyak(a, tgt, p[t_off:], 0.005, dev, file=(raw, ring))
        assert ws[:4].view(torch.int32)[0].item() == 0                   # the rendezvous never timed out
        outs.append((g.clone(), p, m, v, tgt, norms.clone(), state.cpu().clone(), ring.cpu().clone()))
    (g0, p0_, m0, v0, t0, n0, s0, r0), (g1, p1, m1, v1, t1, n1, s1, r1) = outs
    assert torch.equal(g0, g1)
    assert s0[0].item() == s1[0].item() == 2.0 and torch.equal(s0[:3], s1[:3])
    assert torch.equal(r0, r1) and r1[0].tolist() == list(range(64)) and r1[1].tolist() == list(range(64))
    if clip > 0:
        torch.testing.assert_close(n1, n0, rtol=1e-5, atol=1e-7)
    for x, y in ((p1, p0_), (m1, m0), (v1, v0), (t1, t0)):
        torch.testing.assert_close(x, y, rtol=2e-6, atol=2e-7)


def sac_state(g, prefix):
    return {k[len(prefix):].replace("__", "."): torch.tensor(g[k]) for k in g.files if k.startswith(prefix)}


def build_sac(H, w_reg, clip, B):
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.algo import TwinSACQ
    from torchrl.env.synth import SynthVecEnv
    D, A = 17, 6
    dev = torch.device(DEV)
    net = dict(hidden_shapes=[H, H], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.ReLU)
    pf = policies.GuassianContPolicy(input_shape=D, output_shape=2 * A, tanh_action=True, **net)
    qf1 = networks.QNet(input_shape=D + A, output_shape=1, **net)
    qf2 = networks.QNet(input_shape=D + A, output_shape=1, **net)
    env = SynthVecEnv(4, device=dev)

    class Stub:
        epoch_frames = 0

    class Log:
        def add_update_info(self, d): pass
        def add_epoch_info(self, *a, **k): pass
        def log(self, *a): pass
        def finish(self): pass
    return pf, qf1, qf2, dict(plr=3e-4, qlr=1e-3, policy_std_reg_weight=w_reg, policy_mean_reg_weight=w_reg,
                              reparameterization=True, automatic_entropy_tuning=True, env=env, replay_buffer=None,
                              collector=Stub(), logger=Log(), grad_clip=clip, discount=0.99, num_epochs=10,
                              batch_size=B, device=dev, save_dir=None, tau=0.005, use_soft_update=True, opt_times=1), TwinSACQ


@pytest.mark.parametrize("tag", ["h256", "reg"])
def test_twin_sac_q_update_matches_reference(golden, tag):
    g = golden("twin_sac_q")
    B, H, w_reg, clip, steps = g[f"{tag}_args"]
    B, H, steps = int(B), int(H), int(steps)
    pf, qf1, qf2, kw, TwinSACQ = build_sac(H, float(w_reg), float(clip) if clip > 0 else None, B)
    pf.load_state_dict(sac_state(g, f"{tag}_pf0_"))
    qf1.load_state_dict(sac_state(g, f"{tag}_qf10_"))
    qf2.load_state_dict(sac_state(g, f"{tag}_qf20_"))
    agent = TwinSACQ(pf=pf, qf1=qf1, qf2=qf2, **kw)
    for s in range(steps):
        batch = {k: g[f"{tag}_s{s}_batch_{k}"] for k in ("obs", "next_obs", "acts", "rewards", "terminals")}
        torch.manual_seed(100 + s)                                   # the reference's two CPU draws per update
        info = agent.update(batch)
        keys = [str(k) for k in g[f"{tag}_s{s}_info_keys"]]
        assert sorted(info.keys()) == keys
        got = np.array([info[k] for k in keys])
        np.testing.assert_allclose(got, g[f"{tag}_s{s}_info_vals"], rtol=2e-4, atol=5e-5)
    for name, mod in (("pf", pf), ("qf1", qf1), ("qf2", qf2), ("tqf1", agent.target_qf1), ("tqf2", agent.target_qf2)):
        for k, p in mod.state_dict().items():
            err = np.abs(p.cpu().numpy() - g[f"{tag}_{name}1_{k.replace('.', '__')}"]).max()
            assert err < 3e-6, (name, k, err)
    np.testing.assert_allclose(agent.log_alpha.cpu().numpy(), g[f"{tag}_log_alpha"], atol=1e-6)
    assert float(agent.pf_optimizer.state[pf.seq_append_fcs[0].weight]["exp_avg"].abs().sum()) > 0


def test_graph_replayed_updates_equal_eager_updates(golden, monkeypatch):
    """Updates 3+ of a configuration replay a captured HIP graph (the step count, alpha and its moments live on the
    device): the same launches, so parameters and logged statistics are bit-identical to the eager sequence."""
    g = golden("twin_sac_q")
    B, H = int(g["reg_args"][0]), int(g["reg_args"][1])
    gen = torch.Generator().manual_seed(11)
    batches = [{"obs": torch.randn(B, 17, generator=gen), "next_obs": torch.randn(B, 17, generator=gen),
                "acts": torch.rand(B, 6, generator=gen) * 2 - 1, "rewards": torch.randn(B, 1, generator=gen),
                "terminals": (torch.rand(B, 1, generator=gen) < 0.1).float()} for _ in range(6)]
    results = []
    for no_graph in ("1", "0"):
        monkeypatch.setenv("TRL_NO_GRAPH", no_graph)
        pf, qf1, qf2, kw, TwinSACQ = build_sac(H, 1e-3, 1.0, B)
        pf.load_state_dict(sac_state(g, "reg_pf0_"))
        qf1.load_state_dict(sac_state(g, "reg_qf10_"))
        qf2.load_state_dict(sac_state(g, "reg_qf20_"))
        agent = TwinSACQ(pf=pf, qf1=qf1, qf2=qf2, **kw)
        infos = []
        for s, b in enumerate(batches):
            torch.manual_seed(500 + s)
            infos.append(agent.update(b))
        eng = agent.engine()
        assert len(eng._graphs) == (0 if no_graph == "1" else 1)
        assert eng.step_state.cpu().tolist()[0] == len(batches)
        results.append((infos, eng.flat.cpu().clone(), eng.tflat.cpu().clone(), agent.log_alpha.cpu().clone()))
    (ia, fa, ta, la), (ib, fb, tb, lb) = results
    assert torch.equal(fa, fb) and torch.equal(ta, tb) and torch.equal(la, lb)
    for x, y in zip(ia, ib):
        assert x == y


@pytest.mark.parametrize("B,H,D,A,n,act", [(4096, 256, 17, 6, 2, "relu"), (133, 64, 5, 3, 1, "tanh"), (70, 1024, 3, 8, 2, "none")])
def test_one_launch_policy_gradient_equals_the_layer_gemm_and_the_sampler_backward(B, H, D, A, n, act):
    """`trl_sac_policy_grad_f32` against what it replaces -- linear_bwd_input of the critics' first layer (all D + A
    columns) + rsample_bwd_cols on its action columns -- and determinism."""
    from torchrl_amd import _C
    torch.manual_seed(B + H)
    code = {"relu": _C.ACT_RELU, "tanh": _C.ACT_TANH, "none": _C.ACT_NONE}[act]
    dev = torch.device(DEV)
    dys = [torch.randn(B, H, device=dev) for _ in range(n)]
    ys = [torch.tanh(torch.randn(B, H, device=dev)) for _ in range(n)] if act != "none" else [None] * n
    if act == "relu":
        ys = [y.clamp_min(0.0) for y in ys]
    ws = [torch.randn(H, D + A, device=dev) * 0.1 for _ in range(n)]
    head, eps = torch.randn(B, 2 * A, device=dev), torch.randn(B, A, device=dev)
    head[:, A:] *= 8.0                                               # some log_stds outside the clamp
    actv = torch.tanh(head[:, :A] + head[:, A:].clamp(-20, 2).exp() * eps)
    alpha = torch.tensor([0.37], device=dev)
    dxs = [_C.linear_bwd_input(dy, y, code, w) for dy, y, w in zip(dys, ys, ws)]
    want = _C.rsample_bwd_cols(head, eps, actv, dxs[0], dxs[1] if n == 2 else None, D, alpha, 1.0 / B, 1e-3, 2e-3, True)
    assert _C.sac_policy_grad_ok(dys, ws, A)
    got = _C.sac_policy_grad(head, eps, actv, dys, ys, code, ws, D, alpha, 1.0 / B, 1e-3, 2e-3, True)
    again = _C.sac_policy_grad(head, eps, actv, dys, ys, code, ws, D, alpha, 1.0 / B, 1e-3, 2e-3, True)
    assert torch.equal(got, again)
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() <= 2e-5 * max(scale, 1.0), ((got - want).abs().max().item(), scale)
    # the policy's own head backward riding along: dZ2 = (d_head W3) * act'(H2), against the layer launch it replaces
    # (input-gradient GEMM, gated by the launch below it) and against float64
    w3 = torch.randn(2 * A, H, device=dev) * 0.1
    h2 = torch.tanh(torch.randn(B, H, device=dev))
    if act == "relu":
        h2 = h2.clamp_min(0.0)
    d2, dz = _C.sac_policy_grad(head, eps, actv, dys, ys, code, ws, D, alpha, 1.0 / B, 1e-3, 2e-3, True, head_layer=(w3, h2, code))
    assert torch.equal(d2, got) and dz is not None
    raw = _C.linear_bwd_input(got, None, _C.ACT_NONE, w3)                       # (B, H), ungated
    gate = {"relu": (h2 > 0).float(), "tanh": 1.0 - h2 * h2, "none": torch.ones_like(h2)}[act]
    ref64 = (got.double() @ w3.double()) * gate.double()
    sc = ref64.abs().max().item()
    assert (dz.double() - ref64).abs().max().item() <= 2e-6 * max(sc, 1.0)
    assert (dz - raw * gate).abs().max().item() <= 2e-6 * max(sc, 1.0)
    # shapes that do not fit (hidden width of the policy != the critics'): no fused output, d_head unchanged
    d3, none = _C.sac_policy_grad(head, eps, actv, dys, ys, code, ws, D, alpha, 1.0 / B, 1e-3, 2e-3, True,
                                  head_layer=(torch.randn(2 * A, H + 4, device=dev), torch.randn(B, H + 4, device=dev), code))
    assert none is None and torch.equal(d3, got)


def test_sac_update_with_the_streaming_policy_gradient_equals_the_gemm_path(golden):
    """Whole updates with the one-launch policy gradient and with the layer GEMM + sampler launch it replaces: the same
    numbers up to the summation order of 256-term dot products."""
    g = golden("twin_sac_q")
    B, H = int(g["reg_args"][0]), int(g["reg_args"][1])
    gen = torch.Generator().manual_seed(12)
    batches = [{"obs": torch.randn(B, 17, generator=gen), "next_obs": torch.randn(B, 17, generator=gen),
                "acts": torch.rand(B, 6, generator=gen) * 2 - 1, "rewards": torch.randn(B, 1, generator=gen),
                "terminals": (torch.rand(B, 1, generator=gen) < 0.1).float()} for _ in range(4)]
    flats = []
    for fused in (True, False):
        pf, qf1, qf2, kw, TwinSACQ = build_sac(H, 1e-3, 1.0, B)
        pf.load_state_dict(sac_state(g, "reg_pf0_"))
        qf1.load_state_dict(sac_state(g, "reg_qf10_"))
        qf2.load_state_dict(sac_state(g, "reg_qf20_"))
        agent = TwinSACQ(pf=pf, qf1=qf1, qf2=qf2, **kw)
        agent.engine()._pg_fused = fused
        for s, b in enumerate(batches):
            torch.manual_seed(700 + s)
            agent.update(b)
        flats.append(agent.engine().flat.cpu().clone())
    assert (flats[0] - flats[1]).abs().max().item() < 2e-6
    assert not torch.equal(flats[0], torch.zeros_like(flats[0]))


def test_env_step_and_off_policy_collector_vs_oracle():
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.collector import VecCollector
    from torchrl.env.synth import SynthVecEnv
    from torchrl.replay_buffers import BaseReplayBuffer
    N, T, horizon, max_frames, seed, H = 48, 20, 6, 4, 3, 64
    dev = torch.device(DEV)
    torch.manual_seed(9)
    net = dict(hidden_shapes=[H, H], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.ReLU)
    pf = policies.GuassianContPolicy(input_shape=17, output_shape=12, tanh_action=True, **net)
    pf_p = [p.detach().clone() for wb in __import__("torchrl_amd.ops", fromlist=["x"]).linear_layers(pf) for p in wb]
    # stand-alone env step vs the oracle env
    env0, oenv0 = SynthVecEnv(N, horizon=horizon, device=dev), SynthVecEnvCPU(N, horizon=horizon)
    env0.seed(seed); oenv0.seed(seed)
    np.testing.assert_allclose(env0.reset().cpu().numpy(), oenv0.reset(), atol=3e-6)
    acts = np.tanh(np.random.RandomState(0).randn(N, 6)).astype(np.float32)
    o, r, d, info = env0.step(acts)
    oo, orr, od, oinfo = oenv0.step(acts)
    np.testing.assert_allclose(o.cpu().numpy(), oo, atol=2e-6)
    np.testing.assert_allclose(r.cpu().numpy(), orr, atol=2e-6)
    assert np.array_equal(d.cpu().numpy(), od) and info["time_limit"].shape == (N,)

    env, eval_env = SynthVecEnv(N, horizon=horizon, device=dev), SynthVecEnv(N, horizon=horizon, device=dev)
    env.seed(seed)
    buf = BaseReplayBuffer(N * 8, env_nums=N)                        # ring of 8 rows: 20 steps wrap around
    col = VecCollector(env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=dev, epoch_frames=N * T,
                       max_episode_frames=max_frames, eval_episodes=1)
    oenv = SynthVecEnvCPU(N, horizon=horizon)
    oenv.seed(seed)
    ring = replay.RingOracle(N * 8, env_nums=N)
    ocol = VecCollectorOracle(oenv, ring, pf_p, epoch_frames=N * T, max_episode_frames=max_frames)
    torch.manual_seed(5)
    want = ocol.train_one_epoch()
    torch.manual_seed(5)
    got = col.train_one_epoch()
    for k in ("obs", "next_obs", "acts", "rewards", "terminals", "time_limits"):
        err = np.abs(getattr(buf, "_" + k).cpu().numpy() - ring.data[k]).max()
        assert err < 2e-5, (k, err)
    assert buf._top == ring.top and buf._size == ring.size
    assert abs(got["train_epoch_reward"] - want["train_epoch_reward"]) < 1e-3
    np.testing.assert_allclose(np.array(got["train_rewards"], dtype=np.float64),
                               np.array(want["train_rewards"], dtype=np.float64).reshape(-1), atol=1e-4)
    np.testing.assert_allclose(col.current_ob.cpu().numpy(), ocol.current_ob, atol=2e-5)
    ev = col.eval_one_epoch()
    assert len(ev["eval_rewards"]) == N and ev["eval_traj_length"] == horizon


@pytest.mark.parametrize("tag", ["env_limit", "collector_limit", "wrap"])
def test_off_policy_collector_matches_reference(golden, tag):
    """VecCollector.train_one_epoch against the ring the REFERENCE's collector filled (collector/base.py:176-230 run on
    the CPU twin of the synthetic env, tests/golden/collect_offpolicy.npz): env time-limit resets, the collector's
    max_episode_frames resets, a wrapping ring, logged episode returns, collector state."""
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.collector import VecCollector
    from torchrl.env.synth import SynthVecEnv
    from torchrl.replay_buffers import BaseReplayBuffer
    g = golden("collect_offpolicy")
    N, steps, rows, horizon, max_frames, seed = (int(x) for x in g[f"{tag}_args"])
    dev = torch.device(DEV)
    net = dict(hidden_shapes=[32, 32], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.ReLU)
    pf = policies.GuassianContPolicy(input_shape=17, output_shape=12, tanh_action=True, **net)
    pf.load_state_dict(sac_state(g, f"{tag}_pf_"))
    env, eval_env = SynthVecEnv(N, horizon=horizon, device=dev), SynthVecEnv(N, horizon=horizon, device=dev)
    env.seed(seed)
    buf = BaseReplayBuffer(N * rows, env_nums=N)
    col = VecCollector(env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=dev, epoch_frames=N * steps,
                       max_episode_frames=max_frames, eval_episodes=1)
    torch.manual_seed(seed)                                          # the reference's CPU N(0,1) stream (Q5)
    got = col.train_one_epoch()
    for k in ("obs", "next_obs", "acts", "rewards", "terminals", "time_limits"):
        err = np.abs(getattr(buf, "_" + k).cpu().numpy().reshape(g[f"{tag}_buf_{k}"].shape) - g[f"{tag}_buf_{k}"]).max()
        assert err < 2e-5, (k, err)
    assert (buf._top, buf._size) == tuple(int(x) for x in g[f"{tag}_top_size"])
    assert abs(got["train_epoch_reward"] - float(g[f"{tag}_train_epoch_reward"])) < 1e-3
    np.testing.assert_allclose(np.array(got["train_rewards"], dtype=np.float64).reshape(-1), g[f"{tag}_train_rewards"],
                               atol=1e-4)
    np.testing.assert_allclose(col.current_ob.cpu().numpy(), g[f"{tag}_current_ob"], atol=2e-5)


@pytest.mark.parametrize("tag", ["env_limit", "wrap"])
def test_off_policy_collector_on_normalised_env_matches_reference(golden, tag):
    """VecCollector on a NormObs env against what the REFERENCE collected (tests/golden/collect_offpolicy_norm.npz):
    the ring holds normalised observations, the statistics move every step, the policy sees the raw array after any
    reset (Q14), evaluation runs on a copy of the normaliser without updating it."""
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.collector import VecCollector
    from torchrl.env.base_wrapper import NormObs
    from torchrl.env.synth import SynthVecEnv
    from torchrl.replay_buffers import BaseReplayBuffer
    g = golden("collect_offpolicy_norm")
    N, steps, rows, horizon, max_frames, seed = (int(x) for x in g[f"{tag}_args"])
    dev = torch.device(DEV)
    net = dict(hidden_shapes=[32, 32], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.ReLU)
    pf = policies.GuassianContPolicy(input_shape=17, output_shape=12, tanh_action=True, **net)
    pf.load_state_dict(sac_state(g, f"{tag}_pf_"))
    env = NormObs(SynthVecEnv(N, horizon=horizon, device=dev))
    eval_env = NormObs(SynthVecEnv(N, horizon=horizon, device=dev))
    env.seed(seed)
    eval_env.seed(seed + 1)
    buf = BaseReplayBuffer(N * rows, env_nums=N)
    col = VecCollector(env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=dev, epoch_frames=N * steps,
                       max_episode_frames=max_frames, eval_episodes=1)
    np.testing.assert_allclose(col.current_ob.cpu().numpy(), g[f"{tag}_ob0"], atol=2e-6)
    torch.manual_seed(seed)                                          # the reference's CPU N(0,1) stream (Q5)
    got = col.train_one_epoch()
    for k in ("obs", "next_obs", "acts", "rewards", "terminals", "time_limits"):
        err = np.abs(getattr(buf, "_" + k).cpu().numpy().reshape(g[f"{tag}_buf_{k}"].shape) - g[f"{tag}_buf_{k}"]).max()
        assert err < 3e-5, (k, err)
    assert (buf._top, buf._size) == tuple(int(x) for x in g[f"{tag}_top_size"])
    np.testing.assert_allclose(env._obs_normalizer.state.cpu().numpy(), g[f"{tag}_state1"], rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(col.current_ob.cpu().numpy(), g[f"{tag}_current_ob"], atol=3e-5)
    np.testing.assert_allclose(np.array(got["train_rewards"], dtype=np.float64).reshape(-1), g[f"{tag}_train_rewards"],
                               atol=1e-4)
    state = env._obs_normalizer.state.clone()
    ev = col.eval_one_epoch()
    np.testing.assert_allclose(np.array(ev["eval_rewards"], dtype=np.float64).reshape(-1), g[f"{tag}_eval_rewards"],
                               atol=2e-4)
    assert ev["eval_traj_length"] == float(g[f"{tag}_eval_traj_length"])
    assert torch.equal(env._obs_normalizer.state, state)             # evaluation never updates the statistics


@pytest.mark.parametrize("max_frames", [100, 5])
def test_one_launch_vector_step_equals_the_separate_kernels(monkeypatch, max_frames):
    """trl_synth_collect_step_f32 (sample, store, env.step, bookkeeping, partial reset in one launch) against the eight
    separate launches, and its graph-replayed form (the same entry point with its device-side state: step counter / ring row,
    the 32-row ring wraps under replay): replay rows, env and collector state, epoch reward, finished-episode log and
    evaluation, bit for bit; resets by the env's time limit (horizon 7) or by max_episode_frames (5: no episode ends)."""
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.collector import VecCollector
    from torchrl.env.synth import SynthVecEnv
    from torchrl.replay_buffers import BaseReplayBuffer
    N, dev = 300, torch.device(DEV)
    one_launch = VecCollector._one_launch_step

    def run(separate, eager=False):
        # the separate launches are what normalised / host / frame envs run: forced here by declining the one-launch form
        monkeypatch.setattr(VecCollector, "_one_launch_step", (lambda self, env, nz: False) if separate else one_launch)
        monkeypatch.setenv("TRL_NO_GRAPH", "1" if eager else "0")
        torch.manual_seed(3)
        net = dict(hidden_shapes=[64, 64], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.ReLU)
        pf = policies.GuassianContPolicy(input_shape=17, output_shape=12, tanh_action=True, **net).to(dev)
        env, ev = SynthVecEnv(N, horizon=7, device=dev), SynthVecEnv(N, horizon=7, device=dev)
        env.seed(5)
        buf = BaseReplayBuffer(N * 32, env_nums=N)
        col = VecCollector(env=env, eval_env=ev, pf=pf, replay_buffer=buf, device=dev, epoch_frames=N * 20,
                           max_episode_frames=max_frames, eval_episodes=1, noise_mode="device")
        assert col._one_launch_step(env, None) == (not separate)
        res = [col.train_one_epoch() for _ in range(2)]                      # 40 steps: the 32-row ring wraps
        evl = col.eval_one_epoch()
        state = [getattr(buf, "_" + k).clone() for k in ("obs", "acts", "next_obs", "rewards", "terminals", "time_limits")]
        state += [env.cur_obs.clone(), env.t_env.clone(), env.cur_step.clone(), env.episode_idx.clone(), env.ep_return.clone()]
        return res, evl, state, buf._top, col.global_step
    (ra, ea, sa, ta, ga), (rb, eb, sb, tb, gb) = run(True), run(False)
    rc, ec, sc, tc, gc = run(False, eager=True)      # the same launch with host-side step / row arguments, not replayed
    assert ta == tb == tc and ga == gb == gc
    for x, y, z in zip(sa, sb, sc):
        assert torch.equal(x, y) and torch.equal(x, z)
    assert [r["train_rewards"] for r in rc] == [r["train_rewards"] for r in rb] and ec["eval_rewards"] == eb["eval_rewards"]
    for x, y in zip(ra, rb):
        assert x["train_rewards"] == y["train_rewards"] and (len(x["train_rewards"]) > 0) == (max_frames == 100)
        assert abs(x["train_epoch_reward"] - y["train_epoch_reward"]) < 1e-9 * max(1.0, abs(x["train_epoch_reward"]))
    assert ea["eval_rewards"] == eb["eval_rewards"] and ea["eval_traj_length"] == eb["eval_traj_length"] == 7


def test_sac_trains_through_rlalgo_with_device_noise():
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.algo import TwinSACQ
    from torchrl.collector import VecCollector
    from torchrl.env.synth import SynthVecEnv
    from torchrl.replay_buffers import BaseReplayBuffer
    N, dev = 64, torch.device(DEV)
    net = dict(hidden_shapes=[256, 256], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.ReLU)
    pf = policies.GuassianContPolicy(input_shape=17, output_shape=12, tanh_action=True, **net)
    qf1 = networks.QNet(input_shape=23, output_shape=1, **net)
    qf2 = networks.QNet(input_shape=23, output_shape=1, **net)
    env, eval_env = SynthVecEnv(N, horizon=20, device=dev), SynthVecEnv(N, horizon=20, device=dev)
    buf = BaseReplayBuffer(N * 64, env_nums=N)
    col = VecCollector(env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=dev, epoch_frames=N * 8,
                       max_episode_frames=999, eval_episodes=1, noise_mode="device")
    infos = []

    class Log:
        def add_update_info(self, d): infos.append(d)
        def add_epoch_info(self, *a, **k): pass
        def log(self, *a): pass
        def finish(self): pass
    agent = TwinSACQ(pf=pf, qf1=qf1, qf2=qf2, plr=3e-4, qlr=3e-4, policy_std_reg_weight=0, policy_mean_reg_weight=0,
                     reparameterization=True, automatic_entropy_tuning=True, noise_mode="device", env=env,
                     replay_buffer=buf, collector=col, logger=Log(), discount=0.99, num_epochs=2, batch_size=4 * N,
                     device=dev, save_dir=None, tau=0.005, use_soft_update=True, opt_times=4, pretrain_epochs=1,
                     eval_interval=1)
    w0 = pf.seq_append_fcs[0].weight.detach().clone()
    agent.train()
    assert len(infos) == 8 and all(np.isfinite(list(i.values())).all() for i in infos)
    assert (pf.seq_append_fcs[0].weight - w0).abs().max() > 0 and infos[-1]["Alpha"] < 1.0


@pytest.mark.parametrize("B,D,A", [(777, 17, 6), (64, 2, 1), (130, 32, 8), (4096, 23, 3), (65, 11, 3), (200, 33, 2)])
def test_sampling_launch_equals_the_separate_kernels_at_other_shapes(B, D, A):
    """trl_sac_samples_f32 against rsample_fwd / concat2 / philox_normal, bit for bit, in both noise modes and at ragged
    batch sizes, and its per-wave partial moments against float64 torch sums."""
    from torchrl_amd import _C
    gen = torch.Generator().manual_seed(B + D)
    r = lambda *s: torch.randn(*s, generator=gen).to(DEV)
    head, head2, eps1, eps2, obs, acts, nobs = r(B, 2 * A), r(B, 2 * A), r(B, A), r(B, A), r(B, D), r(B, A), r(B, D)
    head[:, A:] *= 8.0                                                    # some log_std outside [-20, 2]
    parts = (B + 63) // 64
    mom = torch.zeros(parts, 12, dtype=torch.float64, device=DEV)
    new_a, logp, next_a, next_logp, x_sa, x_next, x_new = _C.sac_samples(head, head2, eps1, eps2, obs, acts, nobs, mom_part=mom)
    a1, l1 = _C.rsample_fwd(head, eps1)
    a2, l2 = _C.rsample_fwd(head2, eps2)
    assert torch.equal(new_a, a1) and torch.equal(logp, l1) and torch.equal(next_a, a2) and torch.equal(next_logp, l2)
    assert torch.equal(x_sa, _C.concat2(obs, acts)) and torch.equal(x_next, _C.concat2(nobs, a2))
    assert torch.equal(x_new, _C.concat2(obs, a1))
    ls = head[:, A:].clamp(-20.0, 2.0).double()
    mu = head[:, :A].double()
    lp = l1.double().reshape(-1, 1)
    for p in range(parts):
        sl = slice(64 * p, min(B, 64 * p + 64))
        for j, t in enumerate((ls[sl], lp[sl], mu[sl])):
            want = torch.stack([t.sum(), (t * t).sum(), t.max(), (-t).max()])
            assert torch.allclose(mom[p, 4 * j:4 * j + 4], want, rtol=1e-12, atol=1e-12), (p, j)
    # device noise: update u draws (seed, 2u + 1) and (seed, 2u + 2); eps1 receives the first draw
    state = torch.tensor([3.0, 1.0, 1.0, 0.0], dtype=torch.float64, device=DEV)
    e1 = torch.zeros(B, A, device=DEV)
    outs = _C.sac_samples(head, head2, e1, None, obs, acts, nobs, philox=(state, 77))
    w1 = _C.philox_normal(torch.empty(B, A, device=DEV), 77, 7)
    w2 = _C.philox_normal(torch.empty(B, A, device=DEV), 77, 8)
    assert torch.equal(e1, w1)
    b1, m1 = _C.rsample_fwd(head, w1)
    b2, m2 = _C.rsample_fwd(head2, w2)
    assert torch.equal(outs[0], b1) and torch.equal(outs[1], m1) and torch.equal(outs[2], b2) and torch.equal(outs[3], m2)
    assert torch.equal(outs[4], _C.concat2(obs, acts)) and torch.equal(outs[5], _C.concat2(nobs, b2))
    assert torch.equal(outs[6], _C.concat2(obs, b1))
