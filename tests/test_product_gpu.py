"""End-to-end parity of the product classes (torchrl_amd behind the reference's
API) against what the REFERENCE produced for the same seeds
(tests/golden/collect_epoch.npz): collector -> replay buffer -> GAE -> PPO epoch."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class ListLogger:
    def __init__(self):
        self.infos = []

    def add_update_info(self, d):
        self.infos.append(dict(d))

    def add_epoch_info(self, *a, **k):
        pass

    def log(self, *a):
        pass

    def finish(self):
        pass


def build(g, tag, N, T, horizon, max_frames, B, seed, noise_mode="host"):
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.algo import PPO
    from torchrl.collector.on_policy import VecOnPolicyCollector
    from torchrl.env.synth import SynthVecEnv
    from torchrl.replay_buffers.on_policy import OnPolicyReplayBuffer
    dev = torch.device("cuda:0")
    net = dict(hidden_shapes=[64, 64], append_hidden_shapes=[], base_type=networks.MLPBase,
               activation_func=torch.nn.Tanh)
    pf = policies.GuassianContPolicyBasicBias(input_shape=17, output_shape=6, tanh_action=True, **net)
    vf = networks.Net(input_shape=(17,), output_shape=1, **net)
    if g is not None:
        pf.load_state_dict({k[len(tag) + 5:].replace("__", "."): torch.tensor(g[k]) for k in g.files
                            if k.startswith(tag + "_pf0_")})
        vf.load_state_dict({k[len(tag) + 5:].replace("__", "."): torch.tensor(g[k]) for k in g.files
                            if k.startswith(tag + "_vf0_")})
    env = SynthVecEnv(N, horizon=horizon, device=dev)
    eval_env = SynthVecEnv(N, horizon=horizon, device=dev)
    env.seed(seed)
    buf = OnPolicyReplayBuffer(N * T, env_nums=N, time_limit_filter=True)
    col = VecOnPolicyCollector(vf, env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=dev,
                               train_render=False, epoch_frames=N * T, max_episode_frames=max_frames,
                               eval_episodes=1, noise_mode=noise_mode)
    logger = ListLogger()
    agent = PPO(pf=pf, vf=vf, plr=3e-4, vlr=3e-4, clip_para=0.2, opt_epochs=2, tau=0.95, shuffle=True,
                entropy_coeff=0.005, discount=0.99, num_epochs=10, batch_size=B, gae=True, env=env,
                replay_buffer=buf, collector=col, logger=logger, device=dev, save_dir=None)
    return pf, vf, env, buf, col, agent, logger


@pytest.mark.parametrize("engine", ["fused", "generic"])
@pytest.mark.parametrize("tag", ["small", "surpass", "mixed"])
def test_collect_gae_ppo_epoch_matches_reference(golden, tag, engine, monkeypatch, errlog):
    """engine = generic: the arbitrary-shape minibatch loop (dense-layer GEMMs + trl_ppo_generic_losses_f32) forced
    onto the benchmark shape, against the same reference outputs as the fused kernels."""
    monkeypatch.setenv("TRL_GENERIC_PPO", "1" if engine == "generic" else "0")
    g = golden("collect_epoch")
    N, T, horizon, max_frames, B, seed = (int(x) for x in g[f"{tag}_args"])
    pf, vf, env, buf, col, agent, logger = build(g, tag, N, T, horizon, max_frames, B, seed)
    torch.manual_seed(seed)
    res = col.train_one_epoch()
    for k in ("obs", "next_obs", "acts", "values", "rewards", "terminals", "time_limits"):
        err = np.abs(getattr(buf, "_" + k).cpu().numpy() - g[f"{tag}_buf_{k}"]).max()
        assert err < 1e-5, (k, err)
    assert abs(res["train_epoch_reward"] - float(g[f"{tag}_train_epoch_reward"])) < 1e-3
    np.testing.assert_allclose(np.array(res["train_rewards"], dtype=np.float64), g[f"{tag}_train_rewards"], atol=1e-4)
    assert buf._top == 0 and buf._size == T

    agent.current_epoch = 1
    np.random.seed(seed + 100)
    agent.update_per_epoch()
    # advantages: fp32 scan vs fp64 reference, abs 2e-5 / rel 1e-3
    np.testing.assert_allclose(buf._advs.cpu().numpy(), g[f"{tag}_advs"], rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(buf._estimate_returns.cpu().numpy(), g[f"{tag}_rets"], rtol=1e-3, atol=2e-5)
    keys = [str(k) for k in g[f"{tag}_info_keys"]]
    assert sorted(logger.infos[0].keys()) == keys and len(logger.infos) == len(g[f"{tag}_infos"])
    got = np.array([[i[k] for k in keys] for i in logger.infos])
    # scalar losses / statistics: rel 1e-4 / abs 1e-5; min/max log-probs are O(100)
    want_i = g[f"{tag}_infos"]
    errlog("info scalars: max of |got - want| / (1e-5 + 1e-4 |want|)", (np.abs(got - want_i) / (1e-5 + 1e-4 * np.abs(want_i))).max(), 1.0)
    bad = np.argwhere(np.abs(got - want_i) > 1e-5 + 1e-4 * np.abs(want_i))
    assert len(bad) == 0, [(int(r), keys[c], float(got[r, c]), float(want_i[r, c])) for r, c in bad]
    # post-step parameters: the contract is abs 1e-6 after ONE update (checked in test_kernels_gpu.py); this chain takes
    # len(logger.infos) consecutive Adam steps of 3e-4 each, so round-off differences in the clip coefficient compound
    perr = 0.0
    for prefix, mod in (("pf1_", pf), ("vf1_", vf)):
        for name, p in mod.state_dict().items():
            want = g[f"{tag}_{prefix}{name.replace('.', '__')}"]
            perr = max(perr, np.abs(p.cpu().numpy() - want).max())
    errlog("post-epoch params abs (%d updates)" % len(logger.infos), perr, 1e-6)
    assert perr < 1e-6, perr
    # optimiser state is exposed through the torch optimiser objects
    st = agent.pf_optimizer.state[pf.logstd]
    assert float(st["step"]) == len(logger.infos) and st["exp_avg"].abs().sum() > 0
    assert type(agent.engine()).__name__ == ("_GenericPPO" if engine == "generic" else "_FusedPPO")


def test_chain_error_growth_per_update(golden, errlog):
    """The tightest margin of the suite is the parameter bound after the four chained updates of the `surpass` chain (0.92 of
    1e-6).  This is the same chain taken one update at a time -- PPO.update on the batches of one_iteration, against the CPU
    oracle stepping the same batches -- with the parameter error after EVERY update in the error log, so that a kernel
    change which moves a summation order and trips the bound shows where the error enters."""
    from oracle.ppo import PPOOracle
    g = golden("collect_epoch")
    tag = "surpass"
    N, T, horizon, max_frames, B, seed = (int(x) for x in g[f"{tag}_args"])
    pf, vf, env, buf, col, agent, logger = build(g, tag, N, T, horizon, max_frames, B, seed)
    pf_p = [p.detach().cpu().clone() for p in pf._mlp2_param_list()]
    vf_p = [p.detach().cpu().clone() for p in vf._mlp2_param_list()]
    ls = pf.logstd.detach().cpu().clone()
    torch.manual_seed(seed)
    col.train_one_epoch()
    agent.current_epoch = 1
    agent.process_epoch_samples()
    from torchrl.algo import utils as atu
    atu.update_linear_schedule(agent.pf_optimizer, 1, 10, 3e-4)
    atu.update_linear_schedule(agent.vf_optimizer, 1, 10, 3e-4)
    atu.copy_model_params_from_to(agent.pf, agent.target_pf)
    o = PPOOracle(pf_p, ls, vf_p, plr=3e-4, vlr=3e-4, entropy_coeff=0.005, clip_para=0.2, opt_epochs=2, num_epochs=10,
                  batch_size=B)
    o.pf_opt.lr = o.vf_opt.lr = 3e-4 - 3e-4 * (1 / 10.0)
    np.random.seed(seed + 100)
    k = 0
    for _ in range(2):
        for batch in buf.one_iteration(B, agent.sample_key, True):
            agent.update(batch)
            o.update({key: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for key, v in batch.items()})
            k += 1
            got = torch.cat([p.detach().reshape(-1) for p in pf._mlp2_param_list()] + [pf.logstd.detach().reshape(-1)] +
                            [p.detach().reshape(-1) for p in vf._mlp2_param_list()]).cpu()
            want = torch.cat([p.detach().reshape(-1) for p in o.pf] + [o.logstd.detach().reshape(-1)] +
                             [p.detach().reshape(-1) for p in o.vf])
            err = (got - want).abs().max().item()
            errlog("params abs after update %d of the chain (vs the oracle stepping the same batches)" % k, err, 1e-6)
            assert err < 1e-6, (k, err)
    assert k == len(g[f"{tag}_infos"])


def test_update_entry_point_and_one_iteration(golden):
    """PPO.update(batch) with batches from one_iteration (reference call pattern, on_rl_algo.py:37-40)."""
    g = golden("collect_epoch")
    tag = "small"
    N, T, horizon, max_frames, B, seed = (int(x) for x in g[f"{tag}_args"])
    pf, vf, env, buf, col, agent, logger = build(g, tag, N, T, horizon, max_frames, B, seed)
    torch.manual_seed(seed)
    col.train_one_epoch()
    agent.current_epoch = 1
    agent.process_epoch_samples()
    from torchrl.algo import utils as atu
    atu.update_linear_schedule(agent.pf_optimizer, 1, 10, 3e-4)
    atu.update_linear_schedule(agent.vf_optimizer, 1, 10, 3e-4)
    atu.copy_model_params_from_to(agent.pf, agent.target_pf)
    np.random.seed(seed + 100)
    infos = []
    for _ in range(2):
        for batch in buf.one_iteration(B, agent.sample_key, True):
            assert batch["obs"].shape == (B, 17) and batch["advs"].shape == (B, 1)
            infos.append(agent.update(batch))
    keys = [str(k) for k in g[f"{tag}_info_keys"]]
    got = np.array([[i[k] for k in keys] for i in infos])
    np.testing.assert_allclose(got, g[f"{tag}_infos"], rtol=2e-4, atol=5e-5)
    for name, p in pf.state_dict().items():
        assert np.abs(p.cpu().numpy() - g[f"{tag}_pf1_{name.replace('.', '__')}"]).max() < 2e-6


def test_random_batch_index_stream_and_ring(golden):
    from torchrl.replay_buffers import BaseReplayBuffer
    g = golden("index_streams")
    size, N, B, seed = (int(x) for x in g["ring_args"])
    ring = BaseReplayBuffer(size, env_nums=N)
    np.random.seed(seed)
    for t in range(7):
        ring.add_sample({"obs": g["ring_adds"][t], "rewards": g["ring_rew"][t]})
        assert ring._size == g["ring_sizes"][t] and ring._top == g["ring_tops"][t]
        b = ring.random_batch(B, ["obs", "rewards"])
        got = torch.cat([b["obs"], b["rewards"]], -1).cpu().numpy()
        assert np.array_equal(got, g["ring_batches"][t].astype(np.float32))
    assert np.array_equal(ring._obs.cpu().numpy(), g["ring_obs"].astype(np.float32))
    with pytest.raises(AssertionError, match="dividable"):
        ring.random_batch(B + 1, ["obs"])


def test_device_noise_training_runs_and_eval():
    """Fast mode (device Philox) for a few epochs through RLAlgo.train(): finite losses,
    parameters move, evaluation returns one episode per env."""
    pf, vf, env, buf, col, agent, logger = build(None, "", 64, 16, 40, 1000, 256, 0, noise_mode="device")
    p0 = pf.flat_params().clone()
    agent.num_epochs, agent.eval_interval, agent.save_interval = 3, 1, 100
    agent.train()
    assert len(logger.infos) == 3 * 2 * (64 * 16 // 256)
    assert all(np.isfinite(list(i.values())).all() for i in logger.infos)
    assert (pf.flat_params() - p0).abs().max() > 0
    ev = col.eval_one_epoch()
    assert len(ev["eval_rewards"]) == 64 and ev["eval_traj_length"] == 40


def test_epoch_result_read_back_on_first_access_equals_the_immediate_one(monkeypatch):
    """train_one_epoch returns a mapping that is read back when first looked at (header + head of the episode log copied
    behind the rollout in stream order): same values as the immediate read-back, also when the update -- or the next
    rollout, which resolves a result nobody looked at -- is launched in between, and when more episodes ended than the
    speculative copy holds."""
    def run(eager, rows=None):
        torch.manual_seed(0)                                                # same initial networks in every run
        pf, vf, env, buf, col, agent, logger = build(None, "", 64, 16, 5, 1000, 256, 3, noise_mode="device")
        col.eager_epoch_result = eager
        if rows is not None:
            col.SPECULATIVE_ROWS = rows
        out = []
        for epoch in range(3):
            res = col.train_one_epoch()
            assert isinstance(res, dict) == eager
            np.random.seed(epoch)
            agent.current_epoch = epoch
            agent.update_per_epoch()                                        # launched before the result is looked at
            out.append((list(res["train_rewards"]), res["train_epoch_reward"], list(col.train_rews)))
        first = col.train_one_epoch()
        second = col.train_one_epoch()                                      # resolves `first` before reusing the log
        out.append((list(first["train_rewards"]), first["train_epoch_reward"], None))
        out.append((list(second["train_rewards"]), second["train_epoch_reward"], list(col.train_rews)))
        return out, pf.flat_params().cpu().clone()
    (want, pw), (got, pg), (small, ps) = run(True), run(False), run(False, rows=8)
    assert len(want[0][0]) == 64 * 3                                        # horizon 5: three episode ends per env per epoch
    assert want == got == small and torch.equal(pw, pg) and torch.equal(pw, ps)


def test_alternating_epoch_headers_equal_the_memset_path_with_evaluations_in_between():
    """The fused rollout alternates between two {epoch reward, episode count} headers and clears the idle one inside its
    launch; its value pass publishes header + episode log to page-locked memory and leaves the bootstrap value.  Same
    epoch results, evaluation results, buffers and parameters as with a memset in front of every launch, a copy command
    behind it and the bootstrap value from a forward launch -- with evaluations (which use the headers too) in between."""
    def run(plain):
        torch.manual_seed(0)
        pf, vf, env, buf, col, agent, logger = build(None, "", 64, 16, 5, 1000, 256, 3, noise_mode="device")
        if plain:
            launch = col._launch

            def plain_launch(*a, **k):
                col._idle_hdr_clean = False                                 # -> memset of the current header, no swap
                k["publish"] = False                                        # -> copy command behind the launch
                launch(*a, **k)
                buf._boot_fresh = False                                     # -> bootstrap value from vf's forward launch
            col._launch = plain_launch
        out = []
        for epoch in range(4):
            res = col.train_one_epoch()
            np.random.seed(epoch)
            agent.current_epoch = epoch
            agent.update_per_epoch()
            out.append((list(res["train_rewards"]), res["train_epoch_reward"]))
            if epoch in (1, 2):
                ev = col.eval_one_epoch()
                out.append((list(ev["eval_rewards"]), ev["eval_traj_length"]))
        torch.cuda.synchronize()
        return out, pf.flat_params().cpu().clone(), buf._advs.cpu().clone()
    (want, pw, aw), (got, pg, ag) = run(True), run(False)
    assert len(want[0][0]) == 64 * 3
    # The bootstrap value comes from another forward kernel (value pass vs mlp2_forward), so the two runs agree to
    # round-off, not bit for bit, from the first update on; a header mix-up would be an error of the size of the values.
    assert want[0] == got[0]                                                # nothing has been updated yet: identical
    for (wl, ws), (gl, gs) in zip(want, got):
        assert len(wl) == len(gl)
        np.testing.assert_allclose(gl, wl, rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(gs, ws, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(ag, aw, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(pg, pw, rtol=1e-5, atol=1e-6)


def test_update_infos_taken_later_equal_the_ones_read_in_place(monkeypatch):
    """PPO.update_per_epoch with a logger that accepts `add_update_infos_later`: the update is launched and its info dicts
    are assembled when asked for -- by the logger, or by the engine's next run before it reuses the host twin of the
    statistics -- with the next rollout already launched in between; same dicts in the same order, same parameters."""
    class Later:
        def __init__(self): self.infos, self.later, self.calls = [], [], 0
        def add_update_info(self, d): self.drain(); self.infos.append(dict(d))
        def add_update_infos_later(self, resolve): self.calls += 1; self.later.append(resolve)
        def drain(self):
            later, self.later = self.later, []
            for r in later:
                self.infos.extend(dict(d) for d in r())
        def add_epoch_info(self, *a, **k): pass
        def log(self, *a): pass

    def run(deferred):
        torch.manual_seed(0)
        pf, vf, env, buf, col, agent, logger = build(None, "", 64, 16, 5, 1000, 256, 3, noise_mode="device")
        agent.eager_update_infos = not deferred
        log = agent.logger = Later()
        for epoch in range(4):
            col.train_one_epoch()
            np.random.seed(epoch)
            agent.current_epoch = epoch
            agent.update_per_epoch()
            if deferred and epoch == 1:
                log.drain()                                                 # the logger asks first ...
        if deferred:                                                        # ... or the engine did, at its next run
            assert log.calls == 4 and len(log.infos) == 2 * 2 * 4 and agent.training_update_num == 4 * 2 * 4
            eng = agent.engine()                                            # (the last run's statistics are still pending)
            assert getattr(eng, "_pending", None) is not None or len(getattr(eng, "_chain_pend", [])) > 0
        log.drain()
        return log.infos, pf.flat_params().cpu().clone()
    (want, pw), (got, pg) = run(False), run(True)
    assert len(want) == 4 * 2 * 4 and want == got and torch.equal(pw, pg)


def test_example_script_runs_unchanged_api(tmp_path):
    """The example mirrors the reference script's wiring through the `torchrl` alias package."""
    cfg = tmp_path / "ppo_small.json"
    import json
    params = json.load(open(os.path.join(REPO, "config", "ppo_synth_halfcheetah.json")))
    params["replay_buffer"]["size"] = 64 * 32
    params["collector"]["epoch_frames"] = 64 * 32
    params["general_setting"].update(num_epochs=2, batch_size=512, eval_interval=1)
    params["ppo"]["opt_epochs"] = 2
    cfg.write_text(json.dumps(params))
    out = subprocess.run([sys.executable, os.path.join(REPO, "examples", "ppo_continuous_vec.py"), "--config", str(cfg),
                          "--vec_env_nums", "64", "--seed", "1", "--log_dir", str(tmp_path / "log"), "--overwrite"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "EPOCH:1" in out.stdout
    assert os.path.exists(tmp_path / "log" / "ppo_small" / "SynthHalfCheetah-v0" / "1" / "model" / "model_pf_finish.pth")


def test_single_env_example_cfg1(tmp_path):
    """SURVEY.md 8(d) cfg 1 (pass / fail): the reference's examples/ppo_continuous.py wiring -- `get_env`,
    `OnPolicyCollectorBase`, `OnPolicyReplayBuffer(size)` -- with config/ppo_halfcheetah.json's hyper-parameters
    (N = 1, T = 2048, batch 64, 10 opt epochs, obs_norm) on the synthetic env id, a few epochs."""
    import json
    params = json.load(open(os.path.join(REPO, "config", "ppo_synth_halfcheetah_single.json")))
    assert (params["replay_buffer"]["size"], params["general_setting"]["batch_size"], params["ppo"]["opt_epochs"],
            params["env"]["obs_norm"]) == (2048, 64, 10, True)
    params["general_setting"].update(num_epochs=3, eval_interval=1)
    cfg = tmp_path / "ppo_single.json"
    cfg.write_text(json.dumps(params))
    out = subprocess.run([sys.executable, os.path.join(REPO, "examples", "ppo_continuous.py"), "--config", str(cfg),
                          "--seed", "3", "--log_dir", str(tmp_path / "log"), "--overwrite"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "EPOCH:2" in out.stdout
    assert "nan" not in out.stdout.lower()
    model_dir = tmp_path / "log" / "ppo_single" / "SynthHalfCheetah-v0" / "3" / "model"
    assert os.path.exists(model_dir / "model_pf_finish.pth")
    assert any(f.startswith("_obs_normalizer") or "normalizer" in f for f in os.listdir(model_dir)), os.listdir(model_dir)


@pytest.mark.parametrize("noise_mode", ["device", "host"])
def test_two_update_chains_match_the_joint_sequence(golden, monkeypatch, noise_mode):
    """One process runs the critic's and the actor's updates of an epoch as two launch sequences on two streams, with the
    next rollout behind the policy chain and its value pass behind the value chain's end event (`_FusedPPO._run_chains`);
    TRL_PPO_CHAINS=joint is the single sequence.  Six epochs -- eager, captured, replayed -- of collect + update must
    give the same rollout buffers every epoch, the same parameters, Adam state and info dicts: any missing dependency
    between the streams (a rollout reading a policy that is still being stepped, a value pass reading a value function
    that is, statistics read too early) shows up as a difference."""
    g = golden("collect_epoch")
    tag = "mixed"
    N, T, horizon, max_frames, B, seed = (int(x) for x in g[f"{tag}_args"])
    results = []
    for chains in ("joint", "two"):
        monkeypatch.setenv("TRL_PPO_CHAINS", chains)
        torch.manual_seed(seed)
        pf, vf, env, buf, col, agent, logger = build(g, tag, N, T, horizon, max_frames, B, seed, noise_mode=noise_mode)
        # hold every value chain back by ~2 ms of device spin: the next rollout then runs (and finishes) while the chain is
        # still reading the previous rollout's observations, and its value pass has to wait for the chain -- at test
        # size the host would otherwise never be fast enough to make the two overlap (a rollout that overwrote the `obs`
        # the chain reads, round 6, only showed at 16 384 envs)
        agent.engine()._test_value_chain_delay = 5_000_000
        later = []                                                       # the updates are launched, not awaited (utils.Logger's protocol)
        logger.add_update_infos_later = later.append
        snaps = []
        for epoch in range(6):
            col.train_one_epoch()
            agent.current_epoch = epoch
            np.random.seed(seed + epoch)
            agent.update_per_epoch()
            snaps.append({k: getattr(buf, "_" + k).clone() for k in ("obs", "acts", "values", "rewards", "advs", "estimate_returns")})
            if chains == "two" and epoch == 2:
                # the stored observations alternate between two tensors, so a steady run replays two captured sets: the
                # shape's first run is eager, the next two capture -- a default bench (3 warm-up iterations) times replays only
                assert len(agent.engine()._chain_graphs) == 2
        for resolve in later:                                            # read in order, after everything was launched
            logger.infos.extend(dict(d) for d in resolve())
        eng = agent.engine()
        assert eng.two_chains == (chains == "two")
        assert bool(getattr(eng, "_chain_graphs", None)) == (chains == "two")       # (captured launch sequences were replayed)
        saved = {k: v.cpu() for k, v in vf.state_dict().items()}         # so does whoever saves it (Net.state_dict)
        v_now = vf(torch.zeros(3, 17, device="cuda:0"))                  # a reader of the value function settles first
        torch.cuda.synchronize()
        assert all(torch.equal(saved[k], v.cpu()) for k, v in vf.state_dict().items())
        assert int(eng.red_ws[:2].view(torch.int32)[1].item()) == eng.step_count == len(logger.infos)
        if chains == "two":
            assert int(eng.red_ws_v[:2].view(torch.int32)[1].item()) == eng.step_count
        results.append((eng.flat.clone(), eng.m.clone(), eng.v.clone(), logger.infos, snaps, v_now.clone()))
    (f0, m0, v0, i0, s0, y0), (f1, m1, v1, i1, s1, y1) = results
    for e, (a, b) in enumerate(zip(s0, s1)):
        for k in a:
            assert torch.equal(a[k], b[k]), (e, k)
    assert torch.equal(f0, f1) and torch.equal(m0, m1) and torch.equal(v0, v1) and torch.equal(y0, y1)
    assert len(i0) == len(i1) and all(a == b for a, b in zip(i0, i1))


def test_graph_replay_matches_eager_launches(golden, monkeypatch):
    """The captured-and-replayed minibatch loop (third and later epochs of a shape) must be bit-identical to
    launching the same kernels one by one: same parameters, same Adam state, same info dicts."""
    g = golden("collect_epoch")
    tag = "mixed"
    N, T, horizon, max_frames, B, seed = (int(x) for x in g[f"{tag}_args"])
    results = []
    for no_graph in ("1", "0"):
        monkeypatch.setenv("TRL_NO_GRAPH", no_graph)
        pf, vf, env, buf, col, agent, logger = build(g, tag, N, T, horizon, max_frames, B, seed, noise_mode="device")
        for epoch in range(4):
            col.train_one_epoch()
            agent.current_epoch = epoch
            np.random.seed(seed + epoch)
            agent.update_per_epoch()
        eng = agent.engine()
        replayed = getattr(eng, "_graph", None) is not None or bool(getattr(eng, "_chain_graphs", None))
        assert replayed == (no_graph == "0")
        assert int(eng.red_ws[:2].view(torch.int32)[1].item()) == eng.step_count == len(logger.infos)
        results.append((eng.flat.clone(), eng.m.clone(), eng.v.clone(), logger.infos))
    (f0, m0, v0, i0), (f1, m1, v1, i1) = results
    assert torch.equal(f0, f1) and torch.equal(m0, m1) and torch.equal(v0, v1)
    assert len(i0) == len(i1) and all(a == b for a, b in zip(i0, i1))
