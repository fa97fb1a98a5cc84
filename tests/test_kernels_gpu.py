"""Parity of every HIP kernel (called through the C ABI) against the CPU oracle
and the reference-generated golden fixtures.  Needs an MI355X."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import nets, philox, replay
from oracle.collector import VecOnPolicyCollectorOracle
from oracle.ppo import PPOOracle, clip_global_norm
from oracle.synth_env import SynthVecEnvCPU, dynamics_matrices

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(x, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(x)).to(dtype).to(DEV).contiguous()


def flat(params, logstd=None):
    parts = [p.detach().reshape(-1).float() for p in params]
    if logstd is not None:
        parts.append(logstd.detach().reshape(-1).float())
    return torch.cat(parts).to(DEV).contiguous()


def regen_gae_inputs(args):
    T, N, seed, p_term, p_tl = int(args[0]), int(args[1]), int(args[2]), args[3], args[4]
    rs = np.random.RandomState(seed)
    r = rs.randn(T, N, 1).astype(np.float32)
    v = rs.randn(T, N, 1).astype(np.float32)
    d = rs.rand(T, N, 1) < p_term
    tl = (rs.rand(T, N, 1) < p_tl) & d
    lv = rs.randn(N, 1).astype(np.float32)
    return r, v, d, tl, lv


# ------------------------------------------------------------------ K4
@pytest.mark.parametrize("tag", ["kat", "small", "ragged", "one", "cfg2"])
def test_gae_and_discount_vs_reference_golden(golden, tag):
    from torchrl_amd import _C
    g = golden("gae")
    if tag == "cfg2":
        r, v, d, tl, lv = regen_gae_inputs(g["cfg2_args"])
        gamma, tau, stride = g["cfg2_args"][5], g["cfg2_args"][6], int(g["cfg2_args"][7])
    else:
        r, v, d, tl, lv = (g[f"{tag}_{k}"] for k in ("rewards", "values", "terminals", "time_limits", "last_value"))
        gamma, tau, stride = (0.99, 0.95, 1) if tag == "kat" else (g[f"{tag}_args"][5], g[f"{tag}_args"][6], 1)
    T, N = r.shape[:2]
    R, V, Dm, TL = dev(r[..., 0]), dev(v[..., 0]), dev(d[..., 0]), dev(tl[..., 0])
    LV = dev(np.asarray(lv).reshape(N))
    for filt in (0, 1):
        adv, ret = torch.empty_like(R), torch.empty_like(R)
        _C.gae(R, V, Dm, TL, LV, adv, ret, gamma, tau, filt)
        # fp32 scan vs fp64 reference: abs 2e-5 / rel 1e-3
        np.testing.assert_allclose(adv.cpu().numpy()[:, ::stride], g[f"{tag}_gae{filt}_advs"][..., 0], rtol=1e-3, atol=2e-5)
        np.testing.assert_allclose(ret.cpu().numpy()[:, ::stride], g[f"{tag}_gae{filt}_rets"][..., 0], rtol=1e-3, atol=2e-5)
        _C.discount_reward(R, V, Dm, TL, LV, adv, ret, gamma, filt)
        np.testing.assert_allclose(adv.cpu().numpy()[:, ::stride], g[f"{tag}_disc{filt}_advs"][..., 0], rtol=1e-3, atol=5e-5)
        np.testing.assert_allclose(ret.cpu().numpy()[:, ::stride], g[f"{tag}_disc{filt}_rets"][..., 0], rtol=1e-3, atol=5e-5)


@pytest.mark.parametrize("T,N", [(700, 33), (257, 1), (3, 4096)])
def test_gae_long_and_wide_vs_oracle(T, N):
    """T > T_CHUNK exercises the cross-chunk carry; last_terminal masking as on_rl_algo.py:27."""
    from torchrl_amd import _C
    rs = np.random.RandomState(T + N)
    r, v = rs.randn(T, N, 1).astype(np.float32), rs.randn(T, N, 1).astype(np.float32)
    d = rs.rand(T, N, 1) < 0.02
    tl = (rs.rand(T, N, 1) < 0.5) & d
    lv = rs.randn(N, 1).astype(np.float32)
    lt = (rs.rand(N, 1) < 0.3)
    want_a, want_r = replay.gae(r, v, d, tl, lv * (1 - lt), 0.99, 0.95, True)
    adv, ret = torch.empty(T, N, device=DEV), torch.empty(T, N, device=DEV)
    _C.gae(dev(r[..., 0]), dev(v[..., 0]), dev(d[..., 0]), dev(tl[..., 0]), dev(lv[:, 0]), adv, ret, 0.99, 0.95, 1,
           last_terminal=dev(lt[:, 0]))
    np.testing.assert_allclose(adv.cpu().numpy(), want_a[..., 0], rtol=1e-3, atol=5e-5)
    np.testing.assert_allclose(ret.cpu().numpy(), want_r[..., 0], rtol=1e-3, atol=5e-5)


def test_gae_rejects_bad_arguments():
    from torchrl_amd import _C
    x = torch.zeros(4, 4, device=DEV)
    with pytest.raises(_C.TrlError, match="time_limits is null"):
        _C.check(_C.lib().trl_gae_f32(x.data_ptr(), x.data_ptr(), x.data_ptr(), None, x.data_ptr(), None,
                                      x.data_ptr(), x.data_ptr(), 4, 4, 0.99, 0.95, 1, None), "trl_gae_f32")
    with pytest.raises(_C.TrlError, match="no CPU path"):
        _C.gae(x.cpu(), x, x, x, x[0], x, x, 0.99, 0.95, 1)


# ------------------------------------------------------------------ K5/K6/K7
def test_gather_rows_bit_exact_and_index_stream(golden):
    from torchrl_amd import _C
    g = golden("index_streams")
    T, N, B, E, seed = (int(x) for x in g["oi_args"])
    cat = np.concatenate([g["oi_obs"], g["oi_acts"], g["oi_advs"]], -1).astype(np.float32)   # (T, N, 6)
    src = dev(cat)
    np.random.seed(seed)
    k = 0
    for _ in range(E):
        order = np.random.permutation(T)
        for pos in range(0, T, B // N):
            idx = torch.as_tensor(order[pos:pos + B // N]).to(DEV)
            out = _C.gather_rows(src, idx).reshape(B, -1).cpu().numpy()
            assert np.array_equal(out, g["oi_batches"][k].astype(np.float32))
            k += 1
    # odd row sizes (scalar path) and uint8 frames
    for shape, dt in (((9, 5, 3), torch.float32), ((7, 3, 1), torch.float32), ((6, 2, 4, 84, 84), torch.uint8),
                      ((5, 3, 7), torch.uint8)):
        s = (torch.rand(shape, device=DEV) * 255).to(dt)
        idx = torch.tensor([shape[0] - 1, 0, 2, 2], device=DEV)
        assert torch.equal(_C.gather_rows(s, idx), s[idx])
    assert _C.gather_rows(src, torch.zeros(0, dtype=torch.int64, device=DEV)).shape[0] == 0


def test_adv_stats_vs_torch():
    from torchrl_amd import _C
    T, N, n_mb, rows = 16, 200, 4, 4
    adv = torch.randn(T, N, device=DEV) * 3 + 10
    idx = torch.randperm(T, device=DEV).reshape(n_mb, rows)
    raw = torch.zeros(n_mb, 4, dtype=torch.float64, device=DEV)
    _C.adv_stats(adv, idx, raw)
    for m in range(n_mb):
        x = adv[idx[m]].double().reshape(-1)
        n = x.numel()
        s, sq, mx, nmn = raw[m].tolist()
        assert abs(s / n - x.mean().item()) < 1e-9
        var = (sq - s * s / n) / (n - 1)
        assert abs(var ** 0.5 - x.std().item()) < 1e-9
        assert mx == x.max().item() and -nmn == x.min().item()


# ------------------------------------------------------------------ MLP forward
@pytest.mark.parametrize("O", [1, 6])
@pytest.mark.parametrize("M", [1, 31, 32, 200, 4096])
def test_mlp2_forward_vs_torch(O, M):
    from torchrl_amd import _C
    gen = torch.Generator().manual_seed(M * 7 + O)
    params = nets.init_mlp(17, [64, 64], O, generator=gen)
    params = [p * (4.0 if i % 2 == 0 else 1.0) for i, p in enumerate(params)]   # larger pre-activations
    x = torch.randn(M, 17, generator=gen) * 2
    want = nets.mlp(x, params, "tanh")
    got = _C.mlp2_forward(flat(params), x.to(DEV), 17, 64, O, _C.ACT_TANH).cpu()
    # fp32 tolerance: different summation order + tanh approximation (abs err < 2e-7 per activation)
    err = (got - want).abs().max().item()
    assert err < 5e-6, err
    want = nets.mlp(x, params, "relu")
    got = _C.mlp2_forward(flat(params), x.to(DEV), 17, 64, O, _C.ACT_RELU).cpu()
    assert (got - want).abs().max().item() < 5e-6


def test_gauss_logp_vs_oracle():
    from torchrl_amd import _C
    B, A = 1000, 6
    mean, acts = torch.randn(B, A) * 0.5, torch.tanh(torch.randn(B, A)) * 0.99
    ls = torch.randn(A) * 0.3 - 1
    want = nets.tanh_normal_log_prob(acts, mean, torch.exp(ls).expand_as(mean)).sum(-1)
    got = _C.gauss_logp(mean.to(DEV), acts.to(DEV), ls.to(DEV), True).cpu()
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2e-5, atol=2e-4)
    want = nets.normal_log_density(acts, mean, torch.exp(ls).expand_as(mean)).sum(-1)
    got = _C.gauss_logp(mean.to(DEV), acts.to(DEV), ls.to(DEV), False).cpu()
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2e-5, atol=2e-4)


# ------------------------------------------------------------------ env reset / rollout
def test_synth_reset_vs_oracle_philox():
    from torchrl_amd import _C
    N, D, seed = 100, 17, 3
    env = SynthVecEnvCPU(N)
    env.seed(seed)
    want0 = env.reset()
    obs = torch.zeros(N, D, device=DEV)
    ti = [torch.zeros(N, dtype=torch.int32, device=DEV) for _ in range(2)]
    ep = torch.full((N,), -1, dtype=torch.int32, device=DEV)
    er = torch.ones(N, device=DEV)
    _C.synth_reset(obs, ti[0], ti[1], ep, er, None, seed * N)
    np.testing.assert_allclose(obs.cpu().numpy(), want0, atol=3e-6)
    assert (ep == 0).all() and (er == 0).all()
    mask = np.zeros(N, dtype=bool)
    mask[::3] = True
    want1 = env.partial_reset(mask)
    _C.synth_reset(obs, ti[0], ti[1], ep, er, torch.as_tensor(mask).to(torch.uint8).to(DEV), seed * N)
    np.testing.assert_allclose(obs.cpu().numpy(), want1, atol=3e-6)
    assert ep.cpu().numpy().tolist() == [1 if m else 0 for m in mask]


def params_from(g, prefix, with_logstd):
    names = sorted(k for k in g.files if k.startswith(prefix))
    base = [k for k in names if "base__seq_fcs" in k]
    head = [k for k in names if "seq_append_fcs" in k]
    order = sorted(base, key=lambda k: (int(k.split("__")[-2]), "bias" in k)) + sorted(head, key=lambda k: "bias" in k)
    ps = [torch.tensor(g[k]) for k in order]
    ls = torch.tensor(g[prefix + "logstd"]) if with_logstd else None
    return ps, ls


class DeviceRollout:
    """Minimal harness around trl_rollout_synth_f32 (the product wrapper lives in torchrl_amd.collector)."""

    def __init__(self, N, rows, pf, ls, vf, horizon, max_frames, seed, discount=0.99):
        from torchrl_amd import _C
        self._C = _C
        self.N, self.rows = N, rows
        self.pf, self.vf = flat(pf, ls), flat(vf)
        A_, B_ = dynamics_matrices()
        self.envA, self.envB = dev(A_), dev(B_)
        self.cur_obs = torch.zeros(N, 17, device=DEV)
        self.t_env = torch.zeros(N, dtype=torch.int32, device=DEV)
        self.cur_step = torch.zeros(N, dtype=torch.int32, device=DEV)
        self.ep_idx = torch.full((N,), -1, dtype=torch.int32, device=DEV)
        self.ep_ret = torch.zeros(N, device=DEV)
        self.seed_base = seed * N
        _C.synth_reset(self.cur_obs, self.t_env, self.cur_step, self.ep_idx, self.ep_ret, None, self.seed_base)
        self.buf = {k: torch.zeros(rows, N, f, device=DEV) for k, f in
                    (("obs", 17), ("next_obs", 17), ("acts", 6), ("values", 1), ("rewards", 1),
                     ("terminals", 1), ("time_limits", 1), ("old_logp", 1))}
        self.epoch_reward = torch.zeros(1, dtype=torch.float64, device=DEV)
        self.ep_count = torch.zeros(1, dtype=torch.int32, device=DEV)
        self.ep_log = torch.zeros(4096, 3, device=DEV)
        self.horizon, self.max_frames, self.discount = horizon, max_frames, discount

    def run(self, n_steps, noise=None, top=0, noise_step0=0, store=True, deterministic=False):
        a = self._C.RolloutArgs()
        a.pf_params, a.vf_params = self.pf.data_ptr(), self.vf.data_ptr()
        a.D, a.H, a.A, a.act, a.tanh_action = 17, 64, 6, self._C.ACT_TANH, 1
        a.env_A, a.env_B = self.envA.data_ptr(), self.envB.data_ptr()
        a.reward_scale, a.horizon, a.env_seed_base = 1.0, self.horizon, self.seed_base
        a.cur_obs, a.t_env, a.cur_step = self.cur_obs.data_ptr(), self.t_env.data_ptr(), self.cur_step.data_ptr()
        a.episode_idx, a.ep_return = self.ep_idx.data_ptr(), self.ep_ret.data_ptr()
        a.noise = noise.data_ptr() if noise is not None else None
        a.noise_step0, a.deterministic = noise_step0, int(deterministic)
        if store:
            for k in ("obs", "next_obs", "acts", "values", "rewards", "terminals", "time_limits", "old_logp"):
                setattr(a, k, self.buf[k].data_ptr())
        a.rows, a.top, a.N, a.n_steps = self.rows, top, self.N, n_steps
        a.max_episode_frames, a.discount = self.max_frames, self.discount
        a.epoch_reward, a.ep_count, a.ep_log = self.epoch_reward.data_ptr(), self.ep_count.data_ptr(), self.ep_log.data_ptr()
        a.ep_cap, a.step0 = 4096, 0
        self._C.rollout(a, torch.device(DEV))
        torch.cuda.synchronize()


@pytest.mark.parametrize("tag", ["small", "surpass", "mixed"])
def test_rollout_vs_reference_golden(golden, tag):
    """Same nets, same env seed, the reference's own N(0,1) draws -> same 7 buffer arrays."""
    g = golden("collect_epoch")
    N, T, horizon, max_frames, B, seed = (int(x) for x in g[f"{tag}_args"])
    pf, ls = params_from(g, f"{tag}_pf0_", True)
    vf, _ = params_from(g, f"{tag}_vf0_", False)
    ro = DeviceRollout(N, T, pf, ls, vf, horizon, max_frames, seed)
    ro.run(T, noise=dev(g[f"{tag}_noise"]))
    for k in ("obs", "next_obs", "acts", "values", "rewards", "terminals", "time_limits"):
        got, want = ro.buf[k].cpu().numpy(), g[f"{tag}_buf_{k}"]
        err = np.abs(got - want).max()
        assert err < 1e-5, (k, err)
    assert g[f"{tag}_buf_terminals"].sum() > 0
    assert abs(ro.epoch_reward.item() - float(g[f"{tag}_train_epoch_reward"])) < 1e-3
    np.testing.assert_allclose(ro.cur_obs.cpu().numpy(), g[f"{tag}_current_ob"], atol=1e-5)
    # finished-episode returns in (step, env) order == reference's train_rewards list
    cnt = int(ro.ep_count.item())
    log = ro.ep_log[:cnt].cpu().numpy()
    log = log[np.lexsort((log[:, 1], log[:, 0]))]
    np.testing.assert_allclose(log[:, 2], g[f"{tag}_train_rewards"], atol=1e-4)
    # cached log pi_old equals the oracle's target-policy log-prob of the stored actions
    obs, acts = torch.tensor(g[f"{tag}_buf_obs"]).float().reshape(-1, 17), torch.tensor(g[f"{tag}_buf_acts"]).float().reshape(-1, 6)
    want_lp = nets.policy_update_terms(obs, acts, pf, ls)["log_prob"].detach().numpy().reshape(T, N, 1)
    np.testing.assert_allclose(ro.buf["old_logp"].cpu().numpy(), want_lp, rtol=1e-4, atol=2e-3)


def test_rollout_device_philox_and_ring_wrap_vs_oracle():
    """Device-generated noise (fast mode) against the oracle fed the same Philox stream;
    two launches, ragged N (partial tile), ring wrap-around."""
    N, T, horizon, max_frames, seed = 45, 12, 5, 7, 4
    gen = torch.Generator().manual_seed(1)
    pf, vf = nets.init_mlp(17, [64, 64], 6, generator=gen), nets.init_mlp(17, [64, 64], 1, generator=gen)
    pf = [p * (3.0 if i == 4 else 1.0) for i, p in enumerate(pf)]
    ls = torch.full((6,), float(np.log(0.125)))
    env = SynthVecEnvCPU(N, horizon=horizon)
    env.seed(seed)
    ring = replay.RingOracle(N * T, env_nums=N, time_limit_filter=True)
    col = VecOnPolicyCollectorOracle(env, ring, pf, ls, vf, epoch_frames=N * T, max_episode_frames=max_frames)
    env_seed = np.int64(seed) * N + np.arange(N)
    noise = np.stack([philox.normal_vector(6, 100 + t, 0, philox.TAG_NOISE, env_seed) for t in range(T)])
    col.train_one_epoch(noise=torch.tensor(noise))
    ro = DeviceRollout(N, T, pf, ls, vf, horizon, max_frames, seed)
    ro.run(5, top=8, noise_step0=100)             # rows 8..11, then wraps to row 0
    ro.run(7, top=1, noise_step0=105)             # rows 1..7
    order = [8, 9, 10, 11, 0, 1, 2, 3, 4, 5, 6, 7]
    for k in ("obs", "next_obs", "acts", "values", "rewards", "terminals", "time_limits"):
        got = ro.buf[k].cpu().numpy()[order]
        err = np.abs(got - ring.data[k]).max()
        assert err < 2e-5, (k, err)


def test_rollout_eval_mode_stores_nothing():
    N = 32
    gen = torch.Generator().manual_seed(2)
    pf, vf = nets.init_mlp(17, [64, 64], 6, generator=gen), nets.init_mlp(17, [64, 64], 1, generator=gen)
    ls = torch.full((6,), float(np.log(0.125)))
    ro = DeviceRollout(N, 4, pf, ls, vf, horizon=6, max_frames=1000, seed=0)
    ro.run(6, store=False, deterministic=True)
    assert all(float(v.abs().sum()) == 0.0 for v in ro.buf.values())
    assert int(ro.ep_count.item()) == N
    # greedy returns == oracle with zero noise
    env = SynthVecEnvCPU(N, horizon=6)
    env.seed(0)
    ring = replay.RingOracle(N * 6, env_nums=N)
    col = VecOnPolicyCollectorOracle(env, ring, pf, ls, vf, epoch_frames=N * 6, max_episode_frames=1000)
    res = col.train_one_epoch(noise=torch.zeros(6, N, 6))
    log = ro.ep_log[:N].cpu().numpy()
    log = log[np.argsort(log[:, 1])]
    np.testing.assert_allclose(log[:, 2], np.array(res["train_rewards"]).reshape(-1), atol=1e-4)


# ------------------------------------------------------------------ K8-K11
def oracle_grads(o, batch):
    """Gradients of the oracle's two losses, without stepping."""
    f32 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32)
    obs, acts = f32(batch["obs"]), f32(batch["acts"])
    advs, old_v, rets = f32(batch["advs"]), f32(batch["values"]), f32(batch["estimate_returns"])
    advs = (advs - advs.mean()) / (advs.std() + 1e-5)
    v = nets.mlp(obs, o.vf, o.act)
    if o.clipped_value_loss:
        v_clip = old_v + (v - old_v).clamp(-o.clip_para, o.clip_para)
        vf_loss = 0.5 * torch.max((v - rets) ** 2, (v_clip - rets) ** 2).mean()
    else:
        vf_loss = ((v - rets) ** 2).mean()
    gv = torch.autograd.grad(vf_loss, o.vf)
    out = nets.policy_update_terms(obs, acts, o.pf, o.logstd, o.act, o.tanh_action)
    with torch.no_grad():
        old = nets.policy_update_terms(obs, acts, o.tpf, o.tlogstd, o.act, o.tanh_action)
    ratio = torch.exp(out["log_prob"] - old["log_prob"])
    pl = -torch.min(torch.clamp(ratio, 1 - o.clip_para, 1 + o.clip_para) * advs, ratio * advs).mean() \
        - o.entropy_coeff * out["ent"].mean()
    gp = torch.autograd.grad(pl, o.pf + [o.logstd])
    return gp, gv, old["log_prob"], pl.item(), vf_loss.item()


class DevicePPO:
    def __init__(self, pf, ls, vf, n_wg=8):
        from torchrl_amd import _C
        self._C = _C
        self.params = torch.cat([flat(pf, ls), flat(vf)]).contiguous()
        self.P_pf, self.P_vf = 5708, 5377
        self.m, self.v = torch.zeros_like(self.params), torch.zeros_like(self.params)
        self.grads = torch.zeros_like(self.params)
        self.n_wg = n_wg
        ps = _C.ppo_partial_stride(17, 64, 6)
        self.partial = torch.zeros(n_wg, ps, device=DEV)
        self.scal = torch.zeros(n_wg, 8, dtype=torch.float64, device=DEV)
        self.info = torch.zeros(24, dtype=torch.float64, device=DEV)
        self.norms = torch.zeros(2, device=DEV)
        self.t = 0

    def grad(self, buf, row_idx, rows_mb, N, adv_raw, n_global, clipv=False, clip=0.2, ent=0.005):
        _C = self._C
        a = _C.PpoBatchArgs()
        for k, name in (("obs", "obs"), ("acts", "acts"), ("advs", "advs"), ("rets", "estimate_returns"),
                        ("old_values", "values"), ("old_logp", "old_logp")):
            setattr(a, k, buf[name].data_ptr())
        a.row_idx = row_idx.data_ptr() if row_idx is not None else None
        a.rows_mb, a.N = rows_mb, N
        a.adv_raw, a.n_global = adv_raw.data_ptr(), float(n_global)
        a.pf_params, a.vf_params = self.params.data_ptr(), self.params[self.P_pf:].data_ptr()
        a.D, a.H, a.A, a.act = 17, 64, 6, _C.ACT_TANH
        a.clip_para, a.entropy_coeff, a.clipped_value_loss, a.tanh_action = clip, ent, int(clipv), 1
        a.partial, a.scal_partial, a.n_wg = self.partial.data_ptr(), self.scal.data_ptr(), self.n_wg
        _C.ppo_minibatch_grad(a, torch.device(DEV))
        _C.ppo_reduce(self.partial, self.scal, self.n_wg, 17, 64, 6, self.grads, self.info)

    def step(self, plr, vlr, max_norm=0.5, scale=1.0):
        _C = self._C
        self.t += 1
        a = _C.AdamArgs()
        a.params, a.grads, a.exp_avg, a.exp_avg_sq = (self.params.data_ptr(), self.grads.data_ptr(),
                                                      self.m.data_ptr(), self.v.data_ptr())
        a.n_groups = 2
        a.group_sizes[0], a.group_sizes[1] = self.P_pf, self.P_vf
        a.group_lr[0], a.group_lr[1] = plr, vlr
        a.max_norm, a.beta1, a.beta2, a.eps = max_norm, 0.9, 0.999, 1e-5
        a.step_count, a.grad_scale, a.norms_out = self.t, scale, self.norms.data_ptr()
        _C.clip_adam(a, torch.device(DEV))


@pytest.mark.parametrize("tag", ["small", "clipv", "mid"])
def test_ppo_update_vs_reference_golden(golden, tag, errlog):
    """One PPO.update on the reference's batch: gradients vs oracle autograd, losses/statistics
    vs the reference info dict, post-step parameters and Adam moments vs the reference."""
    from torchrl_amd import _C
    g = golden("ppo_update")
    if int(g[f"{tag}_args"][1]) != 64:
        pytest.skip("H=%d not instantiated (kernels are built for H=64)" % int(g[f"{tag}_args"][1]))
    B, H, clipv, steps = (int(x) for x in g[f"{tag}_args"])
    pf, ls = params_from(g, f"{tag}_pf0_", True)
    vf, _ = params_from(g, f"{tag}_vf0_", False)
    tpf, tls = params_from(g, f"{tag}_tpf0_", True)
    o = PPOOracle(pf, ls, vf, plr=3e-4, vlr=3e-4, entropy_coeff=0.005, clip_para=0.2,
                  clipped_value_loss=bool(clipv), num_epochs=10)
    o.tpf, o.tlogstd = tpf, tls
    batch = {k: g[f"{tag}_batch_{k}"] for k in ("obs", "acts", "advs", "values", "estimate_returns")}
    gp, gv, old_lp, pl, vl = oracle_grads(o, batch)

    buf = {k: dev(v).reshape(1, B, -1) for k, v in batch.items()}
    buf["old_logp"] = old_lp.detach().to(DEV).reshape(1, B, 1).contiguous()
    raw = torch.zeros(1, 4, dtype=torch.float64, device=DEV)
    _C.adv_stats(buf["advs"].reshape(1, B), torch.zeros(1, 1, dtype=torch.int64, device=DEV), raw)
    d = DevicePPO(pf, ls, vf, n_wg=8)
    d.grad(buf, None, 1, B, raw, B, clipv=bool(clipv))
    got = d.grads.cpu()
    want = torch.cat([x.reshape(-1) for x in gp] + [x.reshape(-1) for x in gv])
    scale = want.abs().max().item()
    err = (got - want).abs().max().item()
    # fp32 vs torch-CPU autograd; the reference batches pair random actions with a sigma=0.125 policy,
    # so log-probs are O(100) and exp(lp - lp_old) amplifies fp32 rounding: rel 1e-4 of the largest entry
    assert err < 1e-4 * max(scale, 1.0) + 1e-7, (err, scale)
    info = d.info.cpu().numpy()
    keys = [str(k) for k in g[f"{tag}_info0_keys"]]
    ref = dict(zip(keys, g[f"{tag}_info0_vals"]))
    ent = float(nets.normal_entropy(torch.exp(ls)).sum())
    assert abs(info[0] / B - 0.005 * ent - ref["Training/policy_loss"]) < 1e-4 * abs(ref["Training/policy_loss"]) + 1e-5
    assert abs(info[7] / B - ref["Training/vf_loss"]) < 1e-4 * abs(ref["Training/vf_loss"]) + 1e-5
    assert abs(info[1] / B - ref["logprob/mean"]) < 1e-4 * abs(ref["logprob/mean"]) + 1e-4
    # extrema of O(100) log-probs: rel 1e-4 / abs 1e-5 like every other scalar
    for name, got_v in (("logprob/max", info[3]), ("logprob/min", -info[4])):
        tol_v = 1e-4 * abs(ref[name]) + 1e-5
        errlog(name, abs(got_v - ref[name]), tol_v)
        assert abs(got_v - ref[name]) < tol_v, (name, got_v, ref[name])
    # ratio extrema like every other scalar of the info dict: rel 1e-4 / abs 1e-5
    for name, got_v in (("ratio/max", info[5]), ("ratio/min", -info[6])):
        tol_v = 1e-4 * abs(ref[name]) + 1e-5
        errlog(name, abs(got_v - ref[name]), tol_v)
        assert abs(got_v - ref[name]) < tol_v, (name, got_v, ref[name])
    # optimiser step: post-step parameters within 1e-6 of the reference
    d.step(3e-4, 3e-4)
    want_pf, want_ls = params_from(g, f"{tag}_pf1_", True)
    want_vf, _ = params_from(g, f"{tag}_vf1_", False)
    want_p = torch.cat([flat(want_pf, want_ls), flat(want_vf)]).cpu()
    perr = (d.params.cpu() - want_p).abs().max().item()
    errlog("post-step params abs (one update)", perr, 1e-6)
    assert perr < 1e-6, perr
    norms = d.norms.cpu().numpy()
    assert abs(norms[0] - ref["grad_norm/pf"]) < 1e-4 * ref["grad_norm/pf"] + 1e-6
    assert abs(norms[1] - ref["grad_norm/vf"]) < 1e-4 * ref["grad_norm/vf"] + 1e-6


def test_ppo_grad_row_gather_ragged_and_workgroup_counts():
    """Fused row gather: N not a multiple of 32, shuffled rows, B not a multiple of 32;
    result must not depend on the workgroup count."""
    from torchrl_amd import _C
    T, N, rows_mb = 6, 21, 3
    gen = torch.Generator().manual_seed(5)
    pf, vf = nets.init_mlp(17, [64, 64], 6, generator=gen), nets.init_mlp(17, [64, 64], 1, generator=gen)
    pf = [p * (20.0 if i == 4 else 1.0) for i, p in enumerate(pf)]
    ls = torch.full((6,), -1.5)
    rs = np.random.RandomState(3)
    full = {"obs": rs.randn(T, N, 17), "acts": np.tanh(rs.randn(T, N, 6)) * 0.97, "advs": rs.randn(T, N, 1) * 2,
            "values": rs.randn(T, N, 1), "estimate_returns": rs.randn(T, N, 1)}
    full = {k: v.astype(np.float32) for k, v in full.items()}
    idx = np.array([4, 0, 5])
    batch = {k: v[idx].reshape(rows_mb * N, -1) for k, v in full.items()}
    o = PPOOracle(pf, ls, vf, entropy_coeff=0.01, clip_para=0.1)
    o.tpf = [p + 0.01 * torch.randn(p.shape, generator=gen) for p in o.tpf]
    gp, gv, _, _, _ = oracle_grads(o, batch)
    with torch.no_grad():
        all_lp = nets.policy_update_terms(torch.tensor(full["obs"]).reshape(-1, 17), torch.tensor(full["acts"]).reshape(-1, 6),
                                          o.tpf, o.tlogstd)["log_prob"].reshape(T, N, 1)
    buf = {k: dev(v) for k, v in full.items()}
    buf["old_logp"] = all_lp.to(DEV).contiguous()
    ridx = torch.as_tensor(idx).to(DEV)
    raw = torch.zeros(1, 4, dtype=torch.float64, device=DEV)
    _C.adv_stats(buf["advs"].reshape(T, N), ridx.reshape(1, -1), raw)
    want = torch.cat([x.reshape(-1) for x in gp] + [x.reshape(-1) for x in gv])
    outs = []
    for n_wg in (2, 8, 64):
        d = DevicePPO(pf, ls, vf, n_wg=n_wg)
        d.grad(buf, ridx, rows_mb, N, raw, rows_mb * N, clip=0.1, ent=0.01)
        outs.append(d.grads.cpu())
        err = (outs[-1] - want).abs().max().item()
        assert err < 2e-5 * max(1.0, want.abs().max().item()), (n_wg, err)
    assert (outs[0] - outs[2]).abs().max().item() < 1e-5


def test_clip_adam_multi_step_vs_oracle():
    from oracle.ppo import AdamState
    gen = torch.Generator().manual_seed(8)
    pf, vf = nets.init_mlp(17, [64, 64], 6, generator=gen), nets.init_mlp(17, [64, 64], 1, generator=gen)
    ls = torch.zeros(6)
    d = DevicePPO(pf, ls, vf)
    p_ref = [d.params[:5708].cpu().clone(), d.params[5708:].cpu().clone()]
    opt = [AdamState([p_ref[0]], 3e-4), AdamState([p_ref[1]], 1e-3)]
    for step in range(5):
        gr = torch.randn(5708 + 5377, generator=gen) * (10.0 if step % 2 else 0.001)
        d.grads.copy_(gr.to(DEV))
        d.step(3e-4, 1e-3, max_norm=0.5, scale=0.5)
        for k, sl in enumerate((slice(0, 5708), slice(5708, None))):
            gk, nrm = clip_global_norm([gr[sl] * 0.5], 0.5)
            opt[k].step([p_ref[k]], gk)
            assert abs(d.norms[k].item() - nrm) < 1e-4 * nrm
    assert (d.params.cpu() - torch.cat(p_ref)).abs().max().item() < 1e-6


@pytest.mark.parametrize("n_mb,rows_mb,N,T", [(40, 32, 2048, 128), (3, 5, 37, 9), (1, 1, 16, 4), (300, 2, 64, 16)])
def test_epoch_prologue_equals_adv_stats_and_does_its_side_jobs(n_mb, rows_mb, N, T):
    """`trl_ppo_epoch_prologue_f64`: sliced minibatches (arrival counters carried over launches) give trl_adv_stats_f64's
    numbers up to the summation order, bit-identically from launch to launch; the zero and copy side jobs are done."""
    from torchrl_amd import _C
    dev = torch.device("cuda:0")
    torch.manual_seed(n_mb + N)
    advs = torch.randn(T, N, device=dev) * 3 + 0.5
    idx = torch.randint(0, T, (n_mb, rows_mb), device=dev)
    want = _C.adv_stats(advs, idx, torch.zeros(n_mb, 4, dtype=torch.float64, device=dev))
    ws = _C.ppo_epoch_prologue_workspace(n_mb, dev)
    junk = torch.full((77,), 3.25, dtype=torch.float64, device=dev)
    src, dst = torch.randn(1234, device=dev), torch.zeros(1234, device=dev)
    outs = []
    for it in range(3):
        raw = torch.full((n_mb, 4), -7.0, dtype=torch.float64, device=dev)
        pinned = torch.arange(10, dtype=torch.int64).pin_memory()
        dev10 = torch.zeros(10, dtype=torch.int64, device=dev)
        _C.ppo_epoch_prologue(advs, idx.cpu().pin_memory() if it == 1 else idx, raw, ws, zero=junk if it == 0 else None,
                              copies=[(dst, src), (dev10, pinned)] if it == 0 else [])
        if it == 0:
            assert torch.equal(dev10.cpu(), pinned)
        outs.append(raw)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    assert torch.equal(junk, torch.zeros_like(junk)) and torch.equal(dst, src)
    got = outs[0]
    assert torch.equal(got[:, 2:], want[:, 2:])                            # max / -min: exact
    torch.testing.assert_close(got[:, :2], want[:, :2], rtol=1e-12, atol=1e-9)
    a64 = advs.double()[idx]                                               # (n_mb, rows_mb, N)
    torch.testing.assert_close(got[:, 0], a64.sum((1, 2)), rtol=1e-12, atol=1e-9)
    torch.testing.assert_close(got[:, 1], (a64 * a64).sum((1, 2)), rtol=1e-12, atol=1e-9)


# ------------------------------------------------------------------ the whole minibatch step as one launch
def _step_fixture(D, A, T, N, seed):
    gen = torch.Generator().manual_seed(seed)
    pf, vf = nets.init_mlp(D, [64, 64], A, generator=gen), nets.init_mlp(D, [64, 64], 1, generator=gen)
    ls = torch.full((A,), -1.0) + 0.1 * torch.randn(A, generator=gen)
    rs = np.random.RandomState(seed)
    full = {"obs": rs.randn(T, N, D), "acts": np.tanh(rs.randn(T, N, A)) * 0.97, "advs": rs.randn(T, N, 1) * 2,
            "values": rs.randn(T, N, 1), "estimate_returns": rs.randn(T, N, 1), "old_logp": rs.randn(T, N, 1) * 0.1 - 3.0}
    return pf, ls, vf, {k: dev(v) for k, v in full.items()}


@pytest.mark.parametrize("D,A,N,n_wg,n_pf,device_state", [
    (17, 6, 64, 2, 0, 0), (17, 6, 64, 8, 3, 1), (17, 6, 48, 64, 0, 1), (17, 6, 21, 6, 0, 0),
    (17, 6, 256, 256, 147, 1), (11, 3, 32, 16, 0, 1), (27, 8, 32, 250, 130, 0)])
def test_one_launch_step_is_the_two_launch_sequence_bit_for_bit(D, A, N, n_wg, n_pf, device_state):
    """trl_ppo_minibatch_step_f32 (gradient, fold, clip, Adam: one launch, workgroups meeting inside it) against
    trl_ppo_minibatch_grad_f32 + trl_ppo_reduce_adam_f32 (ppo.py:67-75, 113-122): parameters, Adam moments, folded
    gradient, the 24 statistics and both norms bit for bit over three consecutive steps -- small grids (one workgroup
    owning many fold jobs), the full 256-workgroup grid, ragged N, runtime-dims tiles, host- and device-side step count."""
    from torchrl_amd import _C
    lib = _C.lib()
    T, rows_mb = 8, 4
    pf, ls, vf, buf = _step_fixture(D, A, T, N, seed=D * 100 + n_wg)
    if n_wg > lib.trl_ppo_step_max_workgroups():
        pytest.skip("%d workgroups are not co-resident on this device" % n_wg)
    P_pf, P_vf = 64 * D + 64 + 64 * 64 + 64 + A * 64 + 2 * A, 64 * D + 64 + 64 * 64 + 64 + 64 + 1
    ps = _C.ppo_partial_stride(D, 64, A)
    idx = torch.as_tensor(np.random.RandomState(1).permutation(T)[:rows_mb].astype(np.int64)).to(DEV)
    raw = torch.zeros(1, 4, dtype=torch.float64, device=DEV)
    _C.adv_stats(buf["advs"].reshape(T, N), idx.reshape(1, -1), raw)
    stream = _C.stream_ptr(torch.device(DEV))
    results = []
    for one_launch in (False, True):
        params = torch.cat([flat(pf, ls), flat(vf)]).contiguous()
        assert params.numel() == P_pf + P_vf
        m, v, grads = torch.zeros_like(params), torch.zeros_like(params), torch.zeros_like(params)
        partial = torch.zeros(n_wg, ps, device=DEV)
        scal = torch.zeros(n_wg, 8, dtype=torch.float64, device=DEV)
        ws = torch.zeros(lib.trl_ppo_step_workspace(D, 64, A), device=DEV)
        ws[4:8].view(torch.float64).fill_(1.0)
        ws[2:4] = torch.tensor([3e-4, 1e-3])
        infos, norms = torch.zeros(3, 24, dtype=torch.float64, device=DEV), torch.zeros(3, 2, device=DEV)
        g = _C.PpoBatchArgs()
        for k, name in (("obs", "obs"), ("acts", "acts"), ("advs", "advs"), ("rets", "estimate_returns"),
                        ("old_values", "values"), ("old_logp", "old_logp")):
            setattr(g, k, buf[name].data_ptr())
        g.row_idx, g.rows_mb, g.N = idx.data_ptr(), rows_mb, N
        g.adv_raw, g.n_global = raw.data_ptr(), float(rows_mb * N)
        g.pf_params, g.vf_params = params.data_ptr(), params.data_ptr() + 4 * P_pf
        g.D, g.H, g.A, g.act = D, 64, A, _C.ACT_TANH
        g.clip_para, g.entropy_coeff, g.clipped_value_loss, g.tanh_action = 0.2, 0.005, 1, 1
        g.partial, g.scal_partial, g.n_wg, g.n_wg_pf = partial.data_ptr(), scal.data_ptr(), n_wg, n_pf
        a = _C.AdamArgs()
        a.params, a.grads, a.exp_avg, a.exp_avg_sq = params.data_ptr(), grads.data_ptr(), m.data_ptr(), v.data_ptr()
        a.n_groups = 2
        a.group_sizes[0], a.group_sizes[1] = P_pf, P_vf
        a.group_lr[0], a.group_lr[1] = 3e-4, 1e-3
        a.max_norm, a.beta1, a.beta2, a.eps, a.grad_scale = 0.5, 0.9, 0.999, 1e-5, 1.0
        a.device_state = device_state
        for k in range(3):
            a.step_count, a.norms_out = k + 1, norms[k].data_ptr()
            if one_launch:
                _C.check(lib.trl_ppo_minibatch_step_f32(C.byref(g), grads.data_ptr(), infos[k].data_ptr(), C.byref(a),
                                                        ws.data_ptr(), stream), "trl_ppo_minibatch_step_f32")
            else:
                _C.check(lib.trl_ppo_minibatch_grad_f32(C.byref(g), stream), "trl_ppo_minibatch_grad_f32")
                _C.check(lib.trl_ppo_reduce_adam_f32(partial.data_ptr(), scal.data_ptr(), n_wg, n_pf, D, 64, A, grads.data_ptr(),
                                                     infos[k].data_ptr(), C.byref(a), ws.data_ptr(), stream),
                         "trl_ppo_reduce_adam_f32")
        torch.cuda.synchronize()
        assert int(ws[:1].view(torch.int32).item()) == 0, "a rendezvous timed out"
        if device_state:
            assert int(ws[:2].view(torch.int32)[1].item()) == 3
        results.append([x.cpu() for x in (params, m, v, grads, infos, norms)])
    assert not torch.equal(results[0][0], torch.cat([flat(pf, ls), flat(vf)]).cpu()), "the steps moved nothing"
    for name, x, y in zip(("params", "exp_avg", "exp_avg_sq", "grads", "info", "norms"), *results):
        assert torch.equal(torch.nan_to_num(x, nan=-7.0), torch.nan_to_num(y, nan=-7.0)), name
    assert float(results[1][4][:, 23].abs().sum()) == 0.0


def test_one_launch_step_refuses_grids_that_cannot_be_resident():
    from torchrl_amd import _C
    lib = _C.lib()
    cap = lib.trl_ppo_step_max_workgroups()
    assert 0 < cap <= 256
    pf, ls, vf, buf = _step_fixture(17, 6, 8, 16, seed=3)
    params = torch.cat([flat(pf, ls), flat(vf)]).contiguous()
    before = params.clone()
    n_wg = cap + 2
    g = _C.PpoBatchArgs()
    for k, name in (("obs", "obs"), ("acts", "acts"), ("advs", "advs"), ("rets", "estimate_returns"),
                    ("old_values", "values"), ("old_logp", "old_logp")):
        setattr(g, k, buf[name].data_ptr())
    raw = torch.zeros(1, 4, dtype=torch.float64, device=DEV)
    partial = torch.zeros(n_wg, _C.ppo_partial_stride(17, 64, 6), device=DEV)
    scal = torch.zeros(n_wg, 8, dtype=torch.float64, device=DEV)
    g.row_idx, g.rows_mb, g.N, g.adv_raw, g.n_global = None, 8, 16, raw.data_ptr(), 128.0
    g.pf_params, g.vf_params = params.data_ptr(), params.data_ptr() + 4 * 5708
    g.D, g.H, g.A, g.act, g.tanh_action = 17, 64, 6, _C.ACT_TANH, 1
    g.partial, g.scal_partial, g.n_wg, g.n_wg_pf = partial.data_ptr(), scal.data_ptr(), n_wg, 0
    m, v, grads = torch.zeros_like(params), torch.zeros_like(params), torch.zeros_like(params)
    a = _C.AdamArgs()
    a.params, a.grads, a.exp_avg, a.exp_avg_sq = params.data_ptr(), grads.data_ptr(), m.data_ptr(), v.data_ptr()
    a.n_groups, a.step_count, a.grad_scale = 2, 1, 1.0
    a.group_sizes[0], a.group_sizes[1] = 5708, 5377
    ws = torch.zeros(lib.trl_ppo_step_workspace(17, 64, 6), device=DEV)
    info = torch.zeros(24, dtype=torch.float64, device=DEV)
    rc = lib.trl_ppo_minibatch_step_f32(C.byref(g), grads.data_ptr(), info.data_ptr(), C.byref(a), ws.data_ptr(),
                                        _C.stream_ptr(torch.device(DEV)))
    assert rc == -2 and b"co-resident" in lib.trl_last_error()      # TRL_EUNSUPPORTED
    torch.cuda.synchronize()
    assert torch.equal(params, before)


@pytest.mark.parametrize("D,A,N,n_wg,n_pf", [(17, 6, 64, 8, 3), (17, 6, 256, 256, 147), (11, 3, 21, 6, 2), (27, 8, 32, 16, 9)])
def test_one_network_per_launch_is_the_joint_sequence_bit_for_bit(D, A, N, n_wg, n_pf):
    """The critic's and the actor's halves of an update as separate launches (trl_ppo_batch_t.n_wg_pf = n_wg / -1,
    trl_ppo_reduce_adam_net_f32, each with its own workspace) against the joint launches of the same split
    (ppo.py:93-122 / 41-91 touch disjoint networks, optimisers and statistics): parameters, Adam moments, folded gradient,
    statistics and norms bit for bit over three steps, whichever half is launched first."""
    from torchrl_amd import _C
    lib = _C.lib()
    T, rows_mb = 8, 4
    pf, ls, vf, buf = _step_fixture(D, A, T, N, seed=D * 7 + n_wg)
    P_pf, P_vf = 64 * D + 64 + 64 * 64 + 64 + A * 64 + 2 * A, 64 * D + 64 + 64 * 64 + 64 + 64 + 1
    ps = _C.ppo_partial_stride(D, 64, A)
    idx = torch.as_tensor(np.random.RandomState(2).permutation(T)[:rows_mb].astype(np.int64)).to(DEV)
    raw = torch.zeros(1, 4, dtype=torch.float64, device=DEV)
    _C.adv_stats(buf["advs"].reshape(T, N), idx.reshape(1, -1), raw)
    stream = _C.stream_ptr(torch.device(DEV))
    results = []
    for mode in ("joint", "pf_first", "vf_first"):
        params = torch.cat([flat(pf, ls), flat(vf)]).contiguous()
        m, v, grads = torch.zeros_like(params), torch.zeros_like(params), torch.zeros_like(params)
        n_ws = lib.trl_ppo_reduce_adam_workspace(D, 64, A)
        wss = [torch.zeros(n_ws, device=DEV) for _ in range(2)]
        for ws in wss:
            ws[4:8].view(torch.float64).fill_(1.0)
            ws[2:4] = torch.tensor([3e-4, 1e-3])
        infos, norms = torch.zeros(3, 24, dtype=torch.float64, device=DEV), torch.zeros(3, 2, device=DEV)
        g = _C.PpoBatchArgs()
        for k, name in (("obs", "obs"), ("acts", "acts"), ("advs", "advs"), ("rets", "estimate_returns"),
                        ("old_values", "values"), ("old_logp", "old_logp")):
            setattr(g, k, buf[name].data_ptr())
        g.row_idx, g.rows_mb, g.N = idx.data_ptr(), rows_mb, N
        g.adv_raw, g.n_global = raw.data_ptr(), float(rows_mb * N)
        g.pf_params, g.vf_params = params.data_ptr(), params.data_ptr() + 4 * P_pf
        g.D, g.H, g.A, g.act = D, 64, A, _C.ACT_TANH
        g.clip_para, g.entropy_coeff, g.clipped_value_loss, g.tanh_action = 0.2, 0.005, 1, 1
        a = _C.AdamArgs()
        a.params, a.grads, a.exp_avg, a.exp_avg_sq = params.data_ptr(), grads.data_ptr(), m.data_ptr(), v.data_ptr()
        a.n_groups = 2
        a.group_sizes[0], a.group_sizes[1] = P_pf, P_vf
        a.max_norm, a.beta1, a.beta2, a.eps, a.grad_scale, a.device_state = 0.5, 0.9, 0.999, 1e-5, 1.0, 1
        parts = [torch.zeros(n_wg, ps, device=DEV) for _ in range(2)]
        scals = [torch.zeros(n_wg, 8, dtype=torch.float64, device=DEV) for _ in range(2)]
        for k in range(3):
            a.norms_out = norms[k].data_ptr()
            if mode == "joint":
                g.partial, g.scal_partial, g.n_wg, g.n_wg_pf = parts[0].data_ptr(), scals[0].data_ptr(), n_wg, n_pf
                _C.check(lib.trl_ppo_minibatch_grad_f32(C.byref(g), stream), "grad")
                _C.check(lib.trl_ppo_reduce_adam_f32(parts[0].data_ptr(), scals[0].data_ptr(), n_wg, n_pf, D, 64, A, grads.data_ptr(),
                                                     infos[k].data_ptr(), C.byref(a), wss[0].data_ptr(), stream), "reduce_adam")
                continue
            for net in ((0, 1) if mode == "pf_first" else (1, 0)):
                g.partial, g.scal_partial = parts[net].data_ptr(), scals[net].data_ptr()
                g.n_wg, g.n_wg_pf = (n_pf, n_pf) if net == 0 else (n_wg - n_pf, -1)
                _C.check(lib.trl_ppo_minibatch_grad_f32(C.byref(g), stream), "grad (one network)")
                _C.check(lib.trl_ppo_reduce_adam_net_f32(parts[net].data_ptr(), scals[net].data_ptr(), g.n_wg, net, D, 64, A,
                                                         grads.data_ptr(), infos[k].data_ptr(), C.byref(a), wss[net].data_ptr(), stream),
                         "reduce_adam_net")
        torch.cuda.synchronize()
        assert all(int(ws[:2].view(torch.int32)[0].item()) == 0 for ws in wss)
        assert int(wss[0][:2].view(torch.int32)[1].item()) == 3 and (mode == "joint" or int(wss[1][:2].view(torch.int32)[1].item()) == 3)
        results.append([x.cpu() for x in (params, m, v, grads, infos, norms)])
    for other in results[1:]:
        for name, x, y in zip(("params", "exp_avg", "exp_avg_sq", "grads", "info", "norms"), results[0], other):
            assert torch.equal(torch.nan_to_num(x, nan=-7.0), torch.nan_to_num(y, nan=-7.0)), name
