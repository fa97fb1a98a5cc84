#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE implementation.

Runs only in the build container (needs /root/reference, which never travels to
the GPU box).  The reference needs `gym`, `toolz` and `cv2`, none of which is
installed, so minimal stand-ins (spaces + wrapper base classes; merge_with;
empty cv2) are written to a temp dir first -- these stubs are this repo's own
code and only exist to let the reference modules import.  What is stored in
the fixtures is DATA ONLY: seeded inputs and the reference's outputs.

    python tests/golden/make_golden.py            # (re)writes the .npz files
    python tests/golden/make_golden.py --check    # regenerates into a scratch dir, compares with the committed ones

Each fixture records torch / numpy versions in ``meta``.
"""
import json
import os
import sys
import tempfile
import textwrap

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

STUBS = {
    "gym/__init__.py": """
        from . import spaces
        class Env: pass
        class Wrapper(Env):
            def __init__(self, env):
                self.env = env
                self.action_space = getattr(env, 'action_space', None)
                self.observation_space = getattr(env, 'observation_space', None)
            def step(self, a): return self.env.step(a)
            def reset(self, **kw): return self.env.reset(**kw)
            def seed(self, s=None): return self.env.seed(s)
            def close(self): return self.env.close()
        class ObservationWrapper(Wrapper):
            def reset(self, **kw): return self.observation(self.env.reset(**kw))
            def step(self, a):
                o, r, d, i = self.env.step(a); return self.observation(o), r, d, i
        class RewardWrapper(Wrapper):
            def step(self, a):
                o, r, d, i = self.env.step(a); return o, self.reward(r), d, i
        class ActionWrapper(Wrapper):
            def step(self, a): return self.env.step(self.action(a))
        def make(*a, **k): raise RuntimeError('no gym here')
    """,
    "gym/spaces.py": """
        import numpy as np
        class Box:
            def __init__(self, low, high, shape=None, dtype=np.float32):
                if shape is None: shape = np.shape(low)
                self.low = np.broadcast_to(np.asarray(low, dtype=dtype), shape)
                self.high = np.broadcast_to(np.asarray(high, dtype=dtype), shape)
                self.shape = tuple(shape)
        class Discrete:
            def __init__(self, n): self.n = n; self.shape = ()
    """,
    "toolz/__init__.py": "",
    "toolz/dicttoolz.py": """
        def merge_with(func, *dicts):
            out = {}
            for d in dicts:
                for k, v in d.items(): out.setdefault(k, []).append(v)
            return {k: func(v) for k, v in out.items()}
    """,
    "cv2.py": "",
}


def install_stubs():
    root = tempfile.mkdtemp(prefix="trl_stubs_")
    for rel, src in STUBS.items():
        path = os.path.join(root, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(textwrap.dedent(src))
    sys.path.insert(0, root)
    sys.path.insert(0, REF)          # reference `torchrl` wins over the repo alias
    sys.path.append(REPO)            # for `oracle` (env used as a duck-typed VecEnv)
    np.bool = bool                   # collector/base.py:242 (numpy >= 1.24)


META = json.dumps({"torch": torch.__version__, "numpy": np.__version__,
                   "reference": "RchalYang/torchrl @ /root/reference"})


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, meta=np.array(META), **arrays)
    print("wrote", path, os.path.getsize(path), "bytes")


class NullLogger:
    def __init__(self): self.infos = []
    def add_update_info(self, d): self.infos.append(dict(d))
    def add_epoch_info(self, *a, **k): pass
    def log(self, *a): pass
    def finish(self): pass


def state_arrays(prefix, module):
    return {prefix + k.replace(".", "__"): v.detach().cpu().numpy().copy()
            for k, v in module.state_dict().items()}


# ------------------------------------------------------------------ cases
def case_gae():
    from torchrl.replay_buffers.on_policy import OnPolicyReplayBuffer
    out = {}

    def run(tag, T, N, seed, p_term, p_tl, gamma, tau, stride=1):
        rs = np.random.RandomState(seed)
        r = rs.randn(T, N, 1).astype(np.float32)
        v = rs.randn(T, N, 1).astype(np.float32)
        d = (rs.rand(T, N, 1) < p_term)
        tl = (rs.rand(T, N, 1) < p_tl) & d
        lv = rs.randn(N, 1).astype(np.float32)
        out[tag + "_args"] = np.array([T, N, seed, p_term, p_tl, gamma, tau, stride], dtype=np.float64)
        if stride == 1:
            out.update({tag + "_rewards": r, tag + "_values": v, tag + "_terminals": d,
                        tag + "_time_limits": tl, tag + "_last_value": lv})
        for filt in (0, 1):
            buf = OnPolicyReplayBuffer(T * N, env_nums=N, time_limit_filter=bool(filt))
            buf._rewards, buf._values = r.astype(np.float64), v.astype(np.float64)
            buf._terminals, buf._time_limits = d.astype(np.float64), tl.astype(np.float64)
            buf.generalized_advantage_estimation(lv.astype(np.float64), gamma, tau)
            out[f"{tag}_gae{filt}_advs"] = buf._advs[:, ::stride]
            out[f"{tag}_gae{filt}_rets"] = buf._estimate_returns[:, ::stride]
            buf.discount_reward(lv.astype(np.float64), gamma)
            out[f"{tag}_disc{filt}_advs"] = buf._advs[:, ::stride]
            out[f"{tag}_disc{filt}_rets"] = buf._estimate_returns[:, ::stride]

    # hand-checkable KAT of SURVEY.md section 8(a)
    buf = OnPolicyReplayBuffer(8, env_nums=2, time_limit_filter=True)
    buf._rewards = np.array([[1, .5], [0, 1], [2, -1], [1, 1]], dtype=np.float64)[..., None]
    buf._values = np.array([[.5, .2], [.4, .1], [.3, 0], [.2, -.1]], dtype=np.float64)[..., None]
    buf._terminals = np.array([[0, 0], [0, 1], [0, 0], [0, 0]], dtype=np.float64)[..., None]
    buf._time_limits = np.array([[0, 0], [0, 0], [1, 0], [0, 0]], dtype=np.float64)[..., None]
    lv = np.array([[.1], [.3]])
    for k in ("rewards", "values", "terminals", "time_limits"):
        out["kat_" + k] = getattr(buf, "_" + k)
    out["kat_last_value"] = lv
    for filt in (1, 0):
        buf.time_limit_filter = bool(filt)
        buf.generalized_advantage_estimation(lv, 0.99, 0.95)
        out[f"kat_gae{filt}_advs"], out[f"kat_gae{filt}_rets"] = buf._advs, buf._estimate_returns
        buf.discount_reward(lv, 0.99)
        out[f"kat_disc{filt}_advs"], out[f"kat_disc{filt}_rets"] = buf._advs, buf._estimate_returns

    run("small", 16, 8, 11, 0.1, 0.5, 0.99, 0.95)
    run("ragged", 37, 5, 12, 0.2, 0.7, 0.9, 0.8)
    run("one", 1, 3, 13, 0.5, 0.5, 0.99, 0.95)
    run("cfg2", 128, 2048, 14, 0.01, 0.5, 0.99, 0.95, stride=16)   # inputs regenerated from seed
    save("gae", **out)


def case_index_streams():
    from torchrl.replay_buffers.on_policy import OnPolicyReplayBuffer
    out = {}
    np.random.seed(0)
    out["perm8_seed0"] = np.random.permutation(8)
    np.random.seed(0)
    out["randint100x4_seed0"] = np.random.randint(0, 100, 4)
    # one_iteration: E passes over T rows (ppo.py:34-37) -> index stream + gathered data
    T, N, B, E = 12, 4, 16, 3
    rs = np.random.RandomState(5)
    buf = OnPolicyReplayBuffer(T * N, env_nums=N)
    for t in range(T):
        buf.add_sample({"obs": rs.randn(N, 3).astype(np.float32), "acts": rs.randn(N, 2).astype(np.float32),
                        "advs": rs.randn(N, 1).astype(np.float32)})
    out["oi_obs"], out["oi_acts"], out["oi_advs"] = buf._obs, buf._acts, buf._advs
    out["oi_args"] = np.array([T, N, B, E, 123])
    np.random.seed(123)
    batches = []
    for _ in range(E):
        for b in buf.one_iteration(B, ["obs", "acts", "advs"], True):
            batches.append(np.concatenate([b["obs"], b["acts"], b["advs"]], -1))
    out["oi_batches"] = np.stack(batches)
    batches = [np.concatenate([b["obs"], b["acts"], b["advs"]], -1)
               for b in buf.one_iteration(B, ["obs", "acts", "advs"], False)]
    out["oi_batches_noshuffle"] = np.stack(batches)
    # ring semantics + random_batch (base.py:19-51): 7 adds into 5 rows
    from torchrl.replay_buffers import BaseReplayBuffer
    N = 3
    ring = BaseReplayBuffer(5 * N + 2, env_nums=N)       # 17 // 3 = 5 rows
    rs = np.random.RandomState(6)
    adds = rs.randn(7, N, 4).astype(np.float32)
    rew = rs.randn(7, N, 1).astype(np.float32)
    sizes, tops = [], []
    np.random.seed(77)
    rb = []
    for t in range(7):
        ring.add_sample({"obs": adds[t], "rewards": rew[t]})
        sizes.append(ring._size); tops.append(ring._top)
        b = ring.random_batch(2 * N, ["obs", "rewards"])
        rb.append(np.concatenate([b["obs"], b["rewards"]], -1))
    out.update(ring_adds=adds, ring_rew=rew, ring_sizes=np.array(sizes), ring_tops=np.array(tops),
               ring_obs=ring._obs, ring_rewards=ring._rewards, ring_batches=np.stack(rb),
               ring_args=np.array([17, N, 2 * N, 77]))
    save("index_streams", **out)


def build_nets(D, A, H, seed):
    import torchrl.policies as policies
    import torchrl.networks as networks
    torch.manual_seed(seed)
    net = dict(hidden_shapes=[H, H], append_hidden_shapes=[],
               base_type=networks.MLPBase, activation_func=torch.nn.Tanh)
    pf = policies.GuassianContPolicyBasicBias(input_shape=D, output_shape=A,
                                              tanh_action=True, **net)
    vf = networks.Net(input_shape=(D,), output_shape=1, **net)
    return pf, vf


def make_ppo(pf, vf, env, buf, collector, logger, **kw):
    from torchrl.algo import PPO
    args = dict(plr=3e-4, vlr=3e-4, clip_para=0.2, opt_epochs=2, tau=0.95, shuffle=True,
                entropy_coeff=0.005, discount=0.99, num_epochs=10, batch_size=32, gae=True,
                env=env, replay_buffer=buf, collector=collector, logger=logger,
                device=torch.device("cpu"), save_dir=tempfile.mkdtemp(prefix="trl_save_"))
    args.update(kw)
    return PPO(pf=pf, vf=vf, **args)


class _StubCollector:
    epoch_frames = 0


def case_ppo_update():
    """PPO.update on a random batch: info dict + post-step params + Adam moments."""
    import gym
    from oracle.synth_env import SynthVecEnvCPU
    out = {}
    for tag, B, H, clipv, steps in (("small", 64, 64, False, 2), ("clipv", 96, 64, True, 1),
                                    ("mid", 2048, 64, False, 1)):
        D, A = 17, 6
        pf, vf = build_nets(D, A, H, seed=3)
        env = SynthVecEnvCPU(4)
        env.action_space = gym.spaces.Box(-1, 1, (A,))
        agent = make_ppo(pf, vf, env, None, _StubCollector(), NullLogger(),
                         clipped_value_loss=clipv)
        agent.current_epoch = 3
        # perturb target so ratio != 1 on the first step
        rs = np.random.RandomState(9)
        with torch.no_grad():
            for p in agent.target_pf.parameters():
                p.add_(torch.as_tensor(rs.randn(*p.shape).astype(np.float32)) * 0.01)
        batch = {"obs": rs.randn(B, D).astype(np.float32),
                 "acts": np.tanh(rs.randn(B, A)).astype(np.float32) * 0.98,
                 "advs": rs.randn(B, 1).astype(np.float32) * 2 + 0.5,
                 "values": rs.randn(B, 1).astype(np.float32),
                 "estimate_returns": rs.randn(B, 1).astype(np.float32)}
        out.update({f"{tag}_batch_{k}": v for k, v in batch.items()})
        out.update(state_arrays(f"{tag}_pf0_", pf))
        out.update(state_arrays(f"{tag}_vf0_", vf))
        out.update(state_arrays(f"{tag}_tpf0_", agent.target_pf))
        out[f"{tag}_args"] = np.array([B, H, int(clipv), steps], dtype=np.int64)
        for s in range(steps):
            info = agent.update(batch)
            out[f"{tag}_info{s}_keys"] = np.array(sorted(info.keys()))
            out[f"{tag}_info{s}_vals"] = np.array([info[k] for k in sorted(info.keys())], dtype=np.float64)
            out.update(state_arrays(f"{tag}_pf{s + 1}_", pf))
            out.update(state_arrays(f"{tag}_vf{s + 1}_", vf))
        for name, opt, mod in (("pf", agent.pf_optimizer, pf), ("vf", agent.vf_optimizer, vf)):
            for (pn, p) in mod.named_parameters():
                st = opt.state[p]
                out[f"{tag}_{name}adam_m_{pn.replace('.', '__')}"] = st["exp_avg"].numpy().copy()
                out[f"{tag}_{name}adam_v_{pn.replace('.', '__')}"] = st["exp_avg_sq"].numpy().copy()
    save("ppo_update", **out)


def case_collect_and_epoch():
    """VecOnPolicyCollector.train_one_epoch on the synthetic env (with resets,
    time-limit bootstrap), then PPO.update_per_epoch: buffers, advs, final params."""
    import gym
    from torchrl.replay_buffers.on_policy import OnPolicyReplayBuffer
    from torchrl.collector.on_policy import VecOnPolicyCollector
    from oracle.synth_env import SynthVecEnvCPU
    out = {}
    for tag, N, T, horizon, max_frames, B, seed in (
            ("small", 8, 16, 6, 1000, 32, 0),        # env time-limit resets every 6 steps
            ("surpass", 8, 16, 1000, 5, 64, 1),      # collector over-length bootstrap every 5
            ("mixed", 16, 24, 7, 5, 96, 2)):
        D, A, H = 17, 6, 64
        pf, vf = build_nets(D, A, H, seed=seed + 20)

        def mk():
            e = SynthVecEnvCPU(N, horizon=horizon)
            e.action_space = gym.spaces.Box(-1, 1, (A,))
            return e
        env, eval_env = mk(), mk()
        env.seed(seed)
        torch.manual_seed(seed)
        np.random.seed(seed)
        buf = OnPolicyReplayBuffer(N * T, env_nums=N, time_limit_filter=True)
        col = VecOnPolicyCollector(vf, env=env, eval_env=eval_env, pf=pf, replay_buffer=buf,
                                   device=torch.device("cpu"), train_render=False,
                                   epoch_frames=N * T, max_episode_frames=max_frames,
                                   eval_episodes=1)
        out.update(state_arrays(f"{tag}_pf0_", pf))
        out.update(state_arrays(f"{tag}_vf0_", vf))
        noise_state = torch.get_rng_state()
        res = col.train_one_epoch()
        # the N(0,1) draws the reference consumed (Q5: == torch.randn(N, A) per step)
        after = torch.get_rng_state()
        torch.set_rng_state(noise_state)
        out[f"{tag}_noise"] = torch.stack([torch.randn(N, A) for _ in range(T)]).numpy()
        assert torch.equal(torch.get_rng_state(), after), "noise stream mismatch"
        for k in ("obs", "next_obs", "acts", "values", "rewards", "terminals", "time_limits"):
            out[f"{tag}_buf_{k}"] = getattr(buf, "_" + k).copy()
        out[f"{tag}_train_epoch_reward"] = np.array(res["train_epoch_reward"])
        out[f"{tag}_train_rewards"] = np.array(res["train_rewards"], dtype=np.float64).reshape(-1)
        out[f"{tag}_current_ob"] = np.asarray(col.current_ob).copy()
        out[f"{tag}_args"] = np.array([N, T, horizon, max_frames, B, seed], dtype=np.int64)

        logger = NullLogger()
        agent = make_ppo(pf, vf, env, buf, col, logger, batch_size=B, opt_epochs=2)
        agent.current_epoch = 1
        np.random.seed(seed + 100)
        agent.update_per_epoch()
        out[f"{tag}_advs"] = buf._advs.copy()
        out[f"{tag}_rets"] = buf._estimate_returns.copy()
        keys = sorted(logger.infos[0].keys())
        out[f"{tag}_info_keys"] = np.array(keys)
        out[f"{tag}_infos"] = np.array([[i[k] for k in keys] for i in logger.infos], dtype=np.float64)
        out.update(state_arrays(f"{tag}_pf1_", pf))
        out.update(state_arrays(f"{tag}_vf1_", vf))
    save("collect_epoch", **out)


def case_collect_offpolicy():
    """VecCollector.train_one_epoch (torchrl/collector/base.py:176-230) with a tanh-Gaussian policy on the synthetic env:
    env time-limit resets, the collector's own max_episode_frames resets, and a ring that wraps -- ring arrays, the
    N(0,1) draws, logged episode returns, collector state."""
    import gym
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.collector.base import VecCollector
    from torchrl.replay_buffers.base import BaseReplayBuffer
    from oracle.synth_env import SynthVecEnvCPU
    out = {}
    for tag, N, steps, rows, horizon, max_frames, seed in (
            ("env_limit", 8, 12, 16, 5, 999, 3),         # env time-limit every 5 steps
            ("collector_limit", 8, 12, 16, 1000, 4, 4),   # collector resets every 4 steps
            ("wrap", 4, 20, 7, 6, 5, 5)):                 # both, and the 7-row ring wraps twice
        D, A, H = 17, 6, 32
        torch.manual_seed(seed + 40)
        net = dict(hidden_shapes=[H, H], append_hidden_shapes=[], base_type=networks.MLPBase,
                   activation_func=torch.nn.ReLU)
        pf = policies.GuassianContPolicy(input_shape=D, output_shape=2 * A, tanh_action=True, **net)

        def mk():
            e = SynthVecEnvCPU(N, horizon=horizon)
            e.action_space = gym.spaces.Box(-1, 1, (A,))
            return e
        env, eval_env = mk(), mk()
        env.seed(seed)
        torch.manual_seed(seed)
        buf = BaseReplayBuffer(N * rows, env_nums=N)
        col = VecCollector(env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=torch.device("cpu"),
                           train_render=False, epoch_frames=N * steps, max_episode_frames=max_frames, eval_episodes=1)
        out.update(state_arrays(f"{tag}_pf_", pf))
        noise_state = torch.get_rng_state()
        res = col.train_one_epoch()
        after = torch.get_rng_state()
        torch.set_rng_state(noise_state)
        out[f"{tag}_noise"] = torch.stack([torch.randn(N, A) for _ in range(steps)]).numpy()
        assert torch.equal(torch.get_rng_state(), after), "noise stream mismatch"
        for k in ("obs", "next_obs", "acts", "rewards", "terminals", "time_limits"):
            out[f"{tag}_buf_{k}"] = getattr(buf, "_" + k).copy()
        out[f"{tag}_top_size"] = np.array([buf._top, buf._size], dtype=np.int64)
        out[f"{tag}_train_epoch_reward"] = np.array(res["train_epoch_reward"])
        out[f"{tag}_train_rewards"] = np.array(res["train_rewards"], dtype=np.float64).reshape(-1)
        out[f"{tag}_current_ob"] = np.asarray(col.current_ob).copy()
        out[f"{tag}_current_step"] = np.asarray(col.current_step).copy()
        out[f"{tag}_args"] = np.array([N, steps, rows, horizon, max_frames, seed], dtype=np.int64)
    save("collect_offpolicy", **out)


def vecenv_script(env, kind, N, steps):
    """The call sequence both the generator and tests/test_host_logic_cpu.py drive a VecEnv through."""
    rs = np.random.RandomState(0)
    env.seed(11)
    env.train()
    rec = {"reset": np.array(env.reset(), copy=True), "obs": [], "rew": [], "done": [], "tl": [], "mask": [], "after": []}
    for t in range(steps):
        acts = rs.uniform(-1, 1, size=(N, 1)) if kind == "pendulum" else rs.randint(0, 2, size=(N,))
        obs, rew, done, infos = env.step(acts)
        rec["obs"].append(np.array(obs, copy=True)); rec["rew"].append(np.array(rew, copy=True))
        rec["done"].append(np.array(done, copy=True)); rec["tl"].append(np.array(infos["time_limit"], copy=True))
        if done.any() or (t % 7 == 3 and t < 30):
            mask = done.reshape(-1) | (np.arange(N) == t % N)
            rec["mask"].append(np.concatenate([[t], mask.astype(np.int64)]))
            rec["after"].append(np.array(env.partial_reset(mask), copy=True))
    return {k: np.stack(v) if isinstance(v, list) else v for k, v in rec.items()}


def case_vecenv():
    """The reference's VecEnv (torchrl/env/vecenv.py:6-78) over this repo's pure-Python single envs: seeding rule
    seed * N + i, per-env action split / squeeze, stacking of obs / rewards / dones, merged infos, partial_reset
    returning the whole array."""
    import importlib.util
    from torchrl.env.vecenv import VecEnv
    spec = importlib.util.spec_from_file_location("_py_envs", os.path.join(REPO, "torchrl_amd", "env", "py_envs.py"))
    py_envs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(py_envs)
    out = {}
    for kind, cls, N, steps in (("pendulum", py_envs.PendulumEnv, 4, 230), ("cartpole", py_envs.CartPoleEnv, 4, 60)):
        rec = vecenv_script(VecEnv(N, cls, ()), kind, N, steps)
        out.update({f"{kind}_{k}": v for k, v in rec.items()})
        out[f"{kind}_args"] = np.array([N, steps], dtype=np.int64)
    save("vecenv", **out)


def case_eval_epoch():
    """VecCollector.eval_one_epoch (torchrl/collector/base.py:232-280): greedy actions, the first episode of every
    eval env, eval_episodes rounds.  (a) the synthetic env with a tanh-Gaussian policy; (b) the reference's VecEnv over
    this repo's pure-Python cart-pole with a greedy Q-network policy, where episodes end at different steps."""
    import importlib.util
    import gym
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.collector.base import VecCollector
    from torchrl.env.vecenv import VecEnv
    from torchrl.replay_buffers.base import BaseReplayBuffer
    from oracle.synth_env import SynthVecEnvCPU
    out = {}
    # (a)
    N, A, D, H, horizon, episodes, seed = 8, 6, 17, 32, 6, 2, 6
    torch.manual_seed(seed + 50)
    net = dict(hidden_shapes=[H, H], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.ReLU)
    pf = policies.GuassianContPolicy(input_shape=D, output_shape=2 * A, tanh_action=True, **net)

    def mk():
        e = SynthVecEnvCPU(N, horizon=horizon)
        e.action_space = gym.spaces.Box(-1, 1, (A,))
        return e
    env, eval_env = mk(), mk()
    env.seed(seed)
    eval_env.seed(seed + 1)
    col = VecCollector(env=env, eval_env=eval_env, pf=pf, replay_buffer=BaseReplayBuffer(N * 4, env_nums=N),
                       device=torch.device("cpu"), train_render=False, epoch_frames=N * 4, max_episode_frames=999,
                       eval_episodes=episodes)
    res = col.eval_one_epoch()
    out.update(state_arrays("synth_pf_", pf))
    out["synth_eval_rewards"] = np.array(res["eval_rewards"], dtype=np.float64).reshape(-1)
    out["synth_eval_traj_length"] = np.array(res["eval_traj_length"], dtype=np.float64)
    out["synth_args"] = np.array([N, horizon, episodes, seed], dtype=np.int64)
    # (b)
    spec = importlib.util.spec_from_file_location("_py_envs", os.path.join(REPO, "torchrl_amd", "env", "py_envs.py"))
    py_envs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(py_envs)
    N, H, episodes, seed = 6, 32, 2, 9
    torch.manual_seed(seed + 50)
    env, eval_env = VecEnv(N, py_envs.CartPoleEnv, ()), VecEnv(N, py_envs.CartPoleEnv, ())
    env.seed(seed)
    eval_env.seed(seed + 1)
    qf = networks.Net(input_shape=4, output_shape=2, hidden_shapes=[H, H], append_hidden_shapes=[],
                      base_type=networks.MLPBase, activation_func=torch.nn.ReLU)
    pf = policies.EpsilonGreedyDQNDiscretePolicy(qf, start_epsilon=1.0, end_epsilon=0.05, decay_frames=1000, action_shape=2)
    col = VecCollector(env=env, eval_env=eval_env, pf=pf, replay_buffer=BaseReplayBuffer(N * 4, env_nums=N),
                       device=torch.device("cpu"), train_render=False, epoch_frames=N * 4, max_episode_frames=999,
                       eval_episodes=episodes)
    res = col.eval_one_epoch()
    out.update(state_arrays("cartpole_qf_", qf))
    out["cartpole_eval_rewards"] = np.array(res["eval_rewards"], dtype=np.float64).reshape(-1)
    out["cartpole_eval_traj_length"] = np.array(res["eval_traj_length"], dtype=np.float64)
    out["cartpole_args"] = np.array([N, H, episodes, seed], dtype=np.int64)
    print("eval golden:", out["synth_eval_rewards"][:4], out["synth_eval_traj_length"], out["cartpole_eval_rewards"],
          out["cartpole_eval_traj_length"])
    save("eval_epoch", **out)


def case_eps_greedy():
    """EpsilonGreedyDQNDiscretePolicy.explore (torchrl/policies/discrete_policies.py:43-67) called the way VecCollector
    does: linear epsilon decay per call (through its end), np.random.rand / randint from the global stream, argmax."""
    import torchrl.networks as networks
    import torchrl.policies as policies
    N, A, D, H, calls, decay = 16, 3, 4, 32, 12, 10
    torch.manual_seed(77)
    qf = networks.Net(input_shape=D, output_shape=A, hidden_shapes=[H, H], append_hidden_shapes=[],
                      base_type=networks.MLPBase, activation_func=torch.nn.ReLU)
    pf = policies.EpsilonGreedyDQNDiscretePolicy(qf, start_epsilon=0.9, end_epsilon=0.15, decay_frames=decay, action_shape=A)
    rs = np.random.RandomState(5)
    obs = rs.randn(calls, N, D).astype(np.float32)
    np.random.seed(21)
    acts, eps = [], []
    for c in range(calls):
        out_ = pf.explore(torch.Tensor(obs[c]).unsqueeze(0))
        acts.append(out_["action"].numpy().copy())
        eps.append(pf.epsilon)
    out = state_arrays("qf_", qf)
    out.update(obs=obs, actions=np.stack(acts).astype(np.int64), epsilon=np.array(eps, dtype=np.float64),
               args=np.array([N, A, D, H, calls, decay], dtype=np.int64))
    greedy = np.stack([qf(torch.Tensor(o)).max(dim=-1, keepdim=True)[1].numpy() for o in obs])
    print("eps golden: random fraction", float((out["actions"] != greedy).mean()), out["actions"].shape)
    save("eps_greedy", **out)


def subproc_script(env, N, steps):
    """The call sequence both the generator and tests/test_oracle_golden.py drive a process-parallel vec env through."""
    rs = np.random.RandomState(3)
    env.train()
    rec = {"reset": np.array(env.reset(), copy=True), "obs": [], "rew": [], "done": [], "tl": [], "after": []}
    for t in range(steps):
        obs, rew, done, infos = env.step(np.tanh(rs.randn(N, 6)).astype(np.float32))
        rec["obs"].append(np.array(obs, copy=True)); rec["rew"].append(np.array(rew, copy=True))
        rec["done"].append(np.array(done, copy=True)); rec["tl"].append(np.array(infos["time_limit"], copy=True))
        mask = done.reshape(-1) | (np.arange(N) == t % N)
        rec["after"].append(np.array(env.partial_reset(mask), copy=True))
    return {k: np.stack(v) if isinstance(v, list) else v for k, v in rec.items()}


def case_subproc_vecenv():
    """The reference's SubProcVecEnv (torchrl/env/subproc_vecenv.py:10-157: spawned workers, pipes, per-env action
    split, stacked results, partial resets) over oracle.synth_env.SynthSingleEnvCPU -- what the CPU baseline's
    process-parallel env (oracle.subproc_env.SubProcVecEnvCPU) has to reproduce.  One (env_func, env_args) pair for
    all envs: the list form trips vecenv.py:19, and the workers ignore the seed command (Q15), so every env has seed 0."""
    from torchrl.env.subproc_vecenv import SubProcVecEnv
    from oracle.synth_env import SynthSingleEnvCPU
    os.environ["PYTHONPATH"] = os.pathsep.join(p for p in sys.path if p)     # for the spawned workers
    N, procs, steps, horizon = 6, 3, 9, 4
    env = SubProcVecEnv(procs, N, SynthSingleEnvCPU, (0, horizon))
    try:
        rec = subproc_script(env, N, steps)
    finally:
        env.close()
    save("subproc_vecenv", args=np.array([N, procs, steps, horizon], dtype=np.int64), **rec)


def case_init():
    """networks.init: basic_init / uniform_init draws under torch.manual_seed (Q9)."""
    out = {}
    pf, vf = build_nets(17, 6, 64, seed=42)
    out.update(state_arrays("pf_", pf))
    out.update(state_arrays("vf_", vf))
    save("net_init", **out)


def case_twin_sac_q():
    """TwinSACQ.update (torchrl/algo/off_policy/twin_sac_q.py:84-220) on random batches: info dicts,
    the two N(0,1) draws per update, post-update pf/qf/target params and log_alpha."""
    import gym
    import torchrl.policies as policies
    import torchrl.networks as networks
    from torchrl.algo import TwinSACQ
    from oracle.synth_env import SynthVecEnvCPU
    out = {}
    for tag, B, H, w_reg, clip, steps in (("h256", 256, 256, 0.0, None, 2), ("reg", 96, 64, 1e-3, 1.0, 2)):
        D, A = 17, 6
        torch.manual_seed(31)
        net = dict(hidden_shapes=[H, H], append_hidden_shapes=[], base_type=networks.MLPBase,
                   activation_func=torch.nn.ReLU)
        pf = policies.GuassianContPolicy(input_shape=D, output_shape=2 * A, tanh_action=True, **net)
        qf1 = networks.QNet(input_shape=D + A, output_shape=1, **net)
        qf2 = networks.QNet(input_shape=D + A, output_shape=1, **net)
        env = SynthVecEnvCPU(4)
        env.action_space = gym.spaces.Box(-1, 1, (A,))
        agent = TwinSACQ(pf=pf, qf1=qf1, qf2=qf2, plr=3e-4, qlr=1e-3, policy_std_reg_weight=w_reg,
                         policy_mean_reg_weight=w_reg, reparameterization=True, automatic_entropy_tuning=True,
                         env=env, replay_buffer=None, collector=_StubCollector(), logger=NullLogger(),
                         grad_clip=clip, discount=0.99, num_epochs=10, batch_size=B, device=torch.device("cpu"),
                         save_dir=tempfile.mkdtemp(prefix="trl_save_"), tau=0.005, use_soft_update=True, opt_times=1)
        for name, mod in (("pf", pf), ("qf1", qf1), ("qf2", qf2)):
            out.update(state_arrays(f"{tag}_{name}0_", mod))
        out[f"{tag}_args"] = np.array([B, H, w_reg, clip if clip else 0.0, steps], dtype=np.float64)
        rs = np.random.RandomState(17)
        for s in range(steps):
            batch = {"obs": rs.randn(B, D).astype(np.float32), "next_obs": rs.randn(B, D).astype(np.float32),
                     "acts": np.tanh(rs.randn(B, A)).astype(np.float32), "rewards": rs.randn(B, 1).astype(np.float32),
                     "terminals": (rs.rand(B, 1) < 0.1).astype(np.float32)}
            out.update({f"{tag}_s{s}_batch_{k}": v for k, v in batch.items()})
            torch.manual_seed(100 + s)
            state = torch.get_rng_state()
            info = agent.update(batch)
            after = torch.get_rng_state()
            torch.set_rng_state(state)
            out[f"{tag}_s{s}_eps1"] = torch.randn(B, A).numpy()
            out[f"{tag}_s{s}_eps2"] = torch.randn(B, A).numpy()
            assert torch.equal(torch.get_rng_state(), after), "noise stream mismatch"
            keys = sorted(info.keys())
            out[f"{tag}_s{s}_info_keys"] = np.array(keys)
            out[f"{tag}_s{s}_info_vals"] = np.array([float(info[k]) for k in keys], dtype=np.float64)
        for name, mod in (("pf", pf), ("qf1", qf1), ("qf2", qf2), ("tqf1", agent.target_qf1), ("tqf2", agent.target_qf2)):
            out.update(state_arrays(f"{tag}_{name}1_", mod))
        out[f"{tag}_log_alpha"] = agent.log_alpha.detach().numpy().copy()
    save("twin_sac_q", **out)


def case_dqn():
    """DQN.update / QRDQN.update / quantile_regression_loss on a small conv Q-net over 84x84x4 frames
    (torchrl/algo/off_policy/dqn.py:38-74, qrdqn.py:22-74, algo/utils.py:5-13)."""
    import gym
    import torchrl.networks as networks
    import torchrl.algo.utils as atu
    from torchrl.algo.off_policy.dqn import DQN
    from torchrl.algo.off_policy.qrdqn import QRDQN
    out = {}
    A = 6
    convs = [[8, [8, 8], [4, 4], [0, 0]], [8, [4, 4], [2, 2], [0, 0]], [16, [3, 3], [1, 1], [0, 0]]]

    class Env:
        action_space = gym.spaces.Discrete(A)

    class Pf:
        epsilon = 0.25
    for tag, cls, Q, B, steps in (("dqn", DQN, 1, 12, 2), ("qrdqn", QRDQN, 20, 10, 2)):
        torch.manual_seed(77)
        qf = networks.Net(output_shape=A * Q, base_type=networks.CNNBase, append_hidden_shapes=[32],
                          activation_func=torch.nn.Tanh, input_shape=(4, 84, 84), hidden_shapes=convs)
        kw = dict(qf=qf, pf=Pf(), qlr=2.5e-4, env=Env(), replay_buffer=None, collector=_StubCollector(),
                  logger=NullLogger(), discount=0.99, num_epochs=10, batch_size=B, device=torch.device("cpu"),
                  save_dir=tempfile.mkdtemp(prefix="trl_save_"), tau=0.005, use_soft_update=True, opt_times=1)
        agent = cls(quantile_num=Q, **kw) if Q > 1 else cls(**kw)
        out.update(state_arrays(f"{tag}_qf0_", qf))
        out[f"{tag}_args"] = np.array([B, Q, A, steps, 23], dtype=np.int64)
        rs = np.random.RandomState(23)
        for s in range(steps):
            obs_u8 = rs.randint(0, 256, size=(B, 4, 84, 84)).astype(np.uint8)
            nobs_u8 = rs.randint(0, 256, size=(B, 4, 84, 84)).astype(np.uint8)
            acts = rs.randint(0, A, size=(B, 1) if Q == 1 else (B,))
            batch = {"obs": obs_u8.astype(np.float32) / 255.0 - 0.5, "next_obs": nobs_u8.astype(np.float32) / 255.0 - 0.5,
                     "acts": acts.astype(np.float32), "rewards": rs.randn(B, 1).astype(np.float32),
                     "terminals": (rs.rand(B, 1) < 0.2).astype(np.float32)}
            # frames are regenerated from RandomState(23) by the tests (same call order), not stored
            out.update({f"{tag}_s{s}_{k}": batch[k] for k in ("acts", "rewards", "terminals")})
            info = agent.update(batch)
            keys = sorted(info.keys())
            out[f"{tag}_s{s}_info_keys"] = np.array(keys)
            out[f"{tag}_s{s}_info_vals"] = np.array([float(info[k]) for k in keys], dtype=np.float64)
        out.update(state_arrays(f"{tag}_qf1_", qf))
        out.update(state_arrays(f"{tag}_tqf1_", agent.target_qf))
    # quantile_regression_loss + gradient, stand-alone (Q = 200 as config/qrdqn.json)
    g = torch.Generator().manual_seed(3)
    src = torch.randn(7, 200, generator=g).requires_grad_(True)
    tgt = torch.randn(7, 200, generator=g) * 2
    coef = torch.Tensor((2 * np.arange(200) + 1) / 400.0).view(1, -1)
    loss = atu.quantile_regression_loss(coef, src, tgt)
    loss.backward()
    out.update(qr_src=src.detach().numpy(), qr_tgt=tgt.numpy(), qr_loss=np.array(loss.item()), qr_grad=src.grad.numpy())
    save("dqn", **out)


def case_a2c_update():
    """A2C.update (a2c.py:45-106) on random batches: info dict + post-step params, two consecutive steps."""
    import gym
    from torchrl.algo import A2C
    from oracle.synth_env import SynthVecEnvCPU
    out = {}
    for tag, B in (("small", 64), ("mid", 2048)):
        D, A, H = 17, 6, 64
        pf, vf = build_nets(D, A, H, seed=11)
        env = SynthVecEnvCPU(4)
        env.action_space = gym.spaces.Box(-1, 1, (A,))
        agent = A2C(pf=pf, vf=vf, plr=3e-4, vlr=1e-3, entropy_coeff=0.01, tau=0.95, shuffle=True, discount=0.99,
                    num_epochs=10, batch_size=B, gae=True, env=env, replay_buffer=None, collector=_StubCollector(),
                    logger=NullLogger(), device=torch.device("cpu"), save_dir=tempfile.mkdtemp(prefix="trl_save_"))
        rs = np.random.RandomState(21)
        batch = {"obs": rs.randn(B, D).astype(np.float32),
                 "acts": np.tanh(rs.randn(B, A)).astype(np.float32) * 0.98,
                 "advs": rs.randn(B, 1).astype(np.float32) * 2 + 0.5,
                 "estimate_returns": rs.randn(B, 1).astype(np.float32)}
        out.update({f"{tag}_batch_{k}": v for k, v in batch.items()})
        out.update(state_arrays(f"{tag}_pf0_", pf))
        out.update(state_arrays(f"{tag}_vf0_", vf))
        for s_ in range(2):
            info = agent.update(batch)
            out[f"{tag}_info{s_}_keys"] = np.array(sorted(info.keys()))
            out[f"{tag}_info{s_}_vals"] = np.array([info[k] for k in sorted(info.keys())], dtype=np.float64)
            out.update(state_arrays(f"{tag}_pf{s_ + 1}_", pf))
            out.update(state_arrays(f"{tag}_vf{s_ + 1}_", vf))
    save("a2c_update", **out)


def case_vmpo_update():
    """VMPO.update (v_mpo.py:57-181) on random batches: info dict, post-step params, eta / alpha; three consecutive steps,
    with `target_pf <- pf` once before the first (what update_per_epoch does), so the KL term is live from step 2 on."""
    import gym
    import torchrl.algo.utils as atu
    from torchrl.algo import VMPO
    from oracle.synth_env import SynthVecEnvCPU
    out = {}
    for tag, B, D, A, H in (("small", 64, 17, 6, 64), ("odd", 97, 11, 3, 32)):
        pf, vf = build_nets(D, A, H, seed=13)
        with torch.no_grad():                                   # away from the +-3e-3 head init, logstd spread over the dims
            pf.seq_append_fcs[-1].weight.mul_(30.0)
            vf.seq_append_fcs[-1].weight.mul_(30.0)
            pf.logstd.copy_(torch.linspace(-1.2, -0.4, A))
        env = SynthVecEnvCPU(4)
        env.action_space = gym.spaces.Box(-1, 1, (A,))
        agent = VMPO(pf=pf, vf=vf, plr=1e-3, vlr=1e-3, opt_epochs=2, eta_eps=0.02, alpha_eps=0.1, tau=0.95, shuffle=True,
                     discount=0.99, num_epochs=10, batch_size=B, gae=True, env=env, replay_buffer=None,
                     collector=_StubCollector(), logger=NullLogger(), device=torch.device("cpu"),
                     save_dir=tempfile.mkdtemp(prefix="trl_save_"))
        atu.copy_model_params_from_to(agent.pf, agent.target_pf)
        out[f"{tag}_args"] = np.array([B, D, A, H])
        out.update(state_arrays(f"{tag}_pf0_", pf))
        out.update(state_arrays(f"{tag}_vf0_", vf))
        rs = np.random.RandomState(31)
        for s_ in range(3):
            batch = {"obs": rs.randn(B, D).astype(np.float32),
                     "acts": np.tanh(rs.randn(B, A)).astype(np.float32) * 0.98,
                     "advs": rs.randn(B, 1).astype(np.float32) * 2 + 0.5,
                     "values": rs.randn(B, 1).astype(np.float32),
                     "estimate_returns": rs.randn(B, 1).astype(np.float32)}
            out.update({f"{tag}_s{s_}_batch_{k}": v for k, v in batch.items()})
            info = agent.update(batch)
            out[f"{tag}_s{s_}_info_keys"] = np.array(sorted(info.keys()))
            out[f"{tag}_s{s_}_info_vals"] = np.array([info[k] for k in sorted(info.keys())], dtype=np.float64)
            out[f"{tag}_s{s_}_eta_alpha"] = np.array([agent.eta.item(), agent.alpha.item()], dtype=np.float64)
        out.update(state_arrays(f"{tag}_pf1_", pf))
        out.update(state_arrays(f"{tag}_vf1_", vf))
    save("vmpo_update", **out)


def case_trpo_update():
    """TRPO.update (trpo.py:154-226: surrogate gradient, conjugate gradient on the KL Hessian, line search) and
    TRPO.update_vf (:228-251) on random batches; policies without tanh squashing, as examples/trpo_continuous_vec.py builds
    them.  Two policy steps with a value step in between."""
    import contextlib
    import io
    import gym
    import torchrl.policies as policies
    import torchrl.networks as networks
    from torchrl.algo import TRPO
    from oracle.synth_env import SynthVecEnvCPU
    out = {}
    for tag, B, D, A, H in (("small", 256, 17, 6, 64), ("odd", 193, 11, 3, 32)):
        torch.manual_seed(17)
        net = dict(hidden_shapes=[H, H], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.Tanh)
        pf = policies.GuassianContPolicyBasicBias(input_shape=D, output_shape=A, **net)
        vf = networks.Net(input_shape=(D,), output_shape=1, **net)
        with torch.no_grad():
            pf.seq_append_fcs[-1].weight.mul_(30.0)
            vf.seq_append_fcs[-1].weight.mul_(30.0)
            pf.logstd.copy_(torch.linspace(-1.0, -0.3, A))
        env = SynthVecEnvCPU(4)
        env.action_space = gym.spaces.Box(-1, 1, (A,))
        agent = TRPO(pf=pf, vf=vf, plr=3e-4, vlr=1e-3, max_kl=0.01, cg_damping=0.1, cg_iters=10, residual_tol=1e-10,
                     entropy_coeff=0.01, shuffle=True, v_opt_times=2, tau=0.95, discount=0.99, num_epochs=10, batch_size=64,
                     gae=True, env=env, replay_buffer=None, collector=_StubCollector(), logger=NullLogger(),
                     device=torch.device("cpu"), save_dir=tempfile.mkdtemp(prefix="trl_save_"))
        out[f"{tag}_args"] = np.array([B, D, A, H])
        out.update(state_arrays(f"{tag}_pf0_", pf))
        out.update(state_arrays(f"{tag}_vf0_", vf))
        rs = np.random.RandomState(41)
        for s_ in range(2):
            obs = rs.randn(B, D).astype(np.float32)
            with torch.no_grad():
                mean, std, _ = pf(torch.tensor(obs))
            batch = {"obs": obs, "acts": (mean + std * torch.tensor(rs.randn(B, A).astype(np.float32))).numpy(),
                     "advs": rs.randn(B, 1).astype(np.float32) * 2 + 0.5,
                     "estimate_returns": rs.randn(B, 1).astype(np.float32)}
            out.update({f"{tag}_s{s_}_batch_{k}": v for k, v in batch.items()})
            with contextlib.redirect_stdout(io.StringIO()):
                info = agent.update(batch)
            out[f"{tag}_s{s_}_info_keys"] = np.array(sorted(info.keys()))
            out[f"{tag}_s{s_}_info_vals"] = np.array([info[k] for k in sorted(info.keys())], dtype=np.float64)
            out.update(state_arrays(f"{tag}_pf{s_ + 1}_", pf))
            vinfo = agent.update_vf({"obs": batch["obs"], "estimate_returns": batch["estimate_returns"]})
            out[f"{tag}_s{s_}_vinfo_keys"] = np.array(sorted(vinfo.keys()))
            out[f"{tag}_s{s_}_vinfo_vals"] = np.array([vinfo[k] for k in sorted(vinfo.keys())], dtype=np.float64)
        out.update(state_arrays(f"{tag}_vf1_", vf))
    save("trpo_update", **out)


def case_ddpg_td3():
    """DDPG.update (ddpg.py:42-110) and TD3.update (td3.py:57-154) on random batches with FixGuassianContPolicy:
    info dicts, the N(0,1) draws TD3 consumes, post-update online and target parameters."""
    import gym
    import torchrl.policies as policies
    import torchrl.networks as networks
    from torchrl.algo import DDPG, TD3
    from oracle.synth_env import SynthVecEnvCPU
    out = {}
    D, A, H, B = 17, 6, 64, 96
    common = dict(replay_buffer=None, collector=_StubCollector(), logger=NullLogger(), discount=0.99, num_epochs=10,
                  batch_size=B, device=torch.device("cpu"), tau=0.005, use_soft_update=True, opt_times=1)
    for tag, clip in (("ddpg", None), ("ddpg_clip", 1.0), ("td3", None), ("td3_clip", 0.5)):
        torch.manual_seed(41)
        net = dict(hidden_shapes=[H, H], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.ReLU)
        pf = policies.FixGuassianContPolicy(input_shape=D, output_shape=A, tanh_action=True, norm_std_explore=0.1, **net)
        qf1 = networks.QNet(input_shape=D + A, output_shape=1, **net)
        qf2 = networks.QNet(input_shape=D + A, output_shape=1, **net)
        env = SynthVecEnvCPU(4)
        env.action_space = gym.spaces.Box(-1, 1, (A,))
        kw = dict(common, env=env, grad_clip=clip, save_dir=tempfile.mkdtemp(prefix="trl_save_"))
        if tag.startswith("ddpg"):
            agent = DDPG(pf=pf, qf=qf1, plr=3e-4, qlr=1e-3, **kw)
            mods = (("pf", pf), ("qf1", qf1))
            tmods = (("tpf", agent.target_pf), ("tqf1", agent.target_qf))
        else:
            agent = TD3(pf=pf, qf1=qf1, qf2=qf2, plr=3e-4, qlr=1e-3, policy_update_delay=2, norm_std_policy=0.2,
                        noise_clip=0.5, **kw)
            mods = (("pf", pf), ("qf1", qf1), ("qf2", qf2))
            tmods = (("tpf", agent.target_pf), ("tqf1", agent.target_qf1), ("tqf2", agent.target_qf2))
        for name, mod in mods:
            out.update(state_arrays(f"{tag}_{name}0_", mod))
        rs = np.random.RandomState(19)
        steps = 3
        for s_ in range(steps):
            batch = {"obs": rs.randn(B, D).astype(np.float32), "next_obs": rs.randn(B, D).astype(np.float32),
                     "acts": np.tanh(rs.randn(B, A)).astype(np.float32), "rewards": rs.randn(B, 1).astype(np.float32),
                     "terminals": (rs.rand(B, 1) < 0.1).astype(np.float32)}
            out.update({f"{tag}_s{s_}_batch_{k}": v for k, v in batch.items()})
            torch.manual_seed(200 + s_)
            state = torch.get_rng_state()
            info = agent.update(batch)
            after = torch.get_rng_state()
            torch.set_rng_state(state)
            if tag.startswith("td3"):                      # explore noise of target_pf, then the smoothing noise
                out[f"{tag}_s{s_}_eps_explore"] = torch.randn(B, A).numpy()
                out[f"{tag}_s{s_}_eps_smooth"] = torch.randn(B, A).numpy()
            assert torch.equal(torch.get_rng_state(), after), "noise stream mismatch"
            keys = sorted(info.keys())
            out[f"{tag}_s{s_}_info_keys"] = np.array(keys)
            out[f"{tag}_s{s_}_info_vals"] = np.array([float(info[k]) for k in keys], dtype=np.float64)
        for name, mod in mods + tmods:
            out.update(state_arrays(f"{tag}_{name}1_", mod))
        out[f"{tag}_args"] = np.array([B, H, clip if clip else 0.0, steps], dtype=np.float64)
    save("ddpg_td3", **out)


def case_obs_norm():
    """Running observation normaliser (env/base_wrapper.py:44-121): Normalizer.update_estimate / filt on a
    sequence of batches, and NormObs wrapped around the synthetic vector env under
    VecOnPolicyCollector.train_one_epoch (captures the partial_reset bypass)."""
    import gym
    from torchrl.env.base_wrapper import Normalizer, NormObs
    from torchrl.replay_buffers.on_policy import OnPolicyReplayBuffer
    from torchrl.collector.on_policy import VecOnPolicyCollector
    from oracle.synth_env import SynthVecEnvCPU
    out = {}
    rs = np.random.RandomState(5)
    D = 17
    nz = Normalizer((D,))
    batches, means, variances, counts, filts = [], [], [], [], []
    for k, n in enumerate((8, 8, 3, 16, 1, 8)):
        x = (rs.randn(n, D) * (1.0 + 0.5 * k) + 0.3 * k).astype(np.float32)
        nz.update_estimate(x)
        batches.append(x); means.append(nz._mean.copy()); variances.append(nz._var.copy()); counts.append(nz._count)
        filts.append(nz.filt(x))
    out["unit_sizes"] = np.array([b.shape[0] for b in batches], dtype=np.int64)
    out["unit_x"] = np.concatenate(batches, axis=0)
    out["unit_mean"] = np.stack(means); out["unit_var"] = np.stack(variances); out["unit_count"] = np.array(counts)
    out["unit_filt"] = np.concatenate(filts, axis=0)

    for tag, N, T, horizon, max_frames, seed in (("flow", 8, 20, 6, 1000, 3), ("flow_surpass", 8, 20, 1000, 7, 4)):
        A, H = 6, 64
        pf, vf = build_nets(D, A, H, seed=seed + 40)

        def mk():
            e = SynthVecEnvCPU(N, horizon=horizon)
            e.action_space = gym.spaces.Box(-1, 1, (A,))
            e.observation_space = gym.spaces.Box(-np.inf, np.inf, (D,))
            return NormObs(e)
        env, eval_env = mk(), mk()
        env.seed(seed)
        torch.manual_seed(seed)
        np.random.seed(seed)
        buf = OnPolicyReplayBuffer(N * T, env_nums=N, time_limit_filter=True)
        col = VecOnPolicyCollector(vf, env=env, eval_env=eval_env, pf=pf, replay_buffer=buf,
                                   device=torch.device("cpu"), train_render=False,
                                   epoch_frames=N * T, max_episode_frames=max_frames, eval_episodes=1)
        out.update(state_arrays(f"{tag}_pf_", pf))
        out.update(state_arrays(f"{tag}_vf_", vf))
        out[f"{tag}_ob0"] = np.asarray(col.current_ob).copy()                 # normalised reset obs
        out[f"{tag}_state0"] = np.concatenate([env._obs_normalizer._mean, env._obs_normalizer._var,
                                               [env._obs_normalizer._count]])
        noise_state = torch.get_rng_state()
        res = col.train_one_epoch()
        torch.set_rng_state(noise_state)
        out[f"{tag}_noise"] = torch.stack([torch.randn(N, A) for _ in range(T)]).numpy()
        for k in ("obs", "next_obs", "acts", "values", "rewards", "terminals", "time_limits"):
            out[f"{tag}_buf_{k}"] = getattr(buf, "_" + k).copy()
        out[f"{tag}_train_epoch_reward"] = np.array(res["train_epoch_reward"])
        out[f"{tag}_train_rewards"] = np.array(res["train_rewards"], dtype=np.float64).reshape(-1)
        out[f"{tag}_current_ob"] = np.asarray(col.current_ob).copy()
        out[f"{tag}_state1"] = np.concatenate([env._obs_normalizer._mean, env._obs_normalizer._var,
                                               [env._obs_normalizer._count]])
        out[f"{tag}_args"] = np.array([N, T, horizon, max_frames, seed], dtype=np.int64)
    save("obs_norm", **out)


def case_frame_dedup():
    """LazyFrames / FrameStack / MemoryEfficientReplayBuffer (env/atari_wrapper.py:142-227,
    replay_buffers/memory_efficient_replay_buffer.py:5-33): the reference classes driven by the single-env collection
    loop of collector/base.py:60-104 over a prepared frame sequence per env -- ring wrap-around, env `done` resets,
    collector over-length resets.  Stored: the source frames and actions (inputs) and, for every replay row of every
    env, the stacks the reference's encode_batchs returns (+ the scalar keys); for the single-env case also the
    reference's own random_batch draws."""
    import gym
    from torchrl.env.atari_wrapper import FrameStack
    from torchrl.replay_buffers.memory_efficient_replay_buffer import MemoryEfficientReplayBuffer
    from oracle.frames import FrameSourceCPU, run_single_env
    np.float = float                                             # memory_efficient_replay_buffer.py:25 (numpy >= 1.24)
    out = {}

    class Src(FrameSourceCPU):
        def __init__(self, frames, horizon):
            super().__init__(frames, horizon)
            self.observation_space = gym.spaces.Box(0, 255, frames.shape[1:], dtype=np.uint8)
            self.observation_space.dtype = np.uint8
            self.action_space = gym.spaces.Discrete(6)

    #            tag       N  rows steps H   W   horizon max_frames seed
    for tag, N, rows, steps, H, W, horizon, max_frames, seed in (
            ("done", 4, 6, 17, 12, 12, 5, 1000, 1), ("surpass", 4, 6, 17, 12, 12, 1000, 3, 2),
            ("mixed", 3, 5, 23, 8, 16, 7, 4, 3), ("single", 1, 9, 31, 12, 12, 6, 5, 5),
            ("single84", 1, 4, 9, 84, 84, 6, 1000, 4)):
        rs = np.random.RandomState(seed)
        n_frames = 2 * steps + 2                                 # more than any env can consume
        frames = rs.randint(0, 256, size=(N, n_frames, 1, H, W)).astype(np.uint8)
        acts = rs.randint(0, 6, size=(steps, N))
        bufs = []
        for n in range(N):
            env = FrameStack(Src(frames[n], horizon), 4)
            buf = MemoryEfficientReplayBuffer(rows)
            run_single_env(env.reset, env.step, buf.add_sample, acts[:, n], max_frames)
            bufs.append(buf)
        allrows = list(range(rows))
        for key in ("obs", "next_obs"):
            arr = np.stack([b.encode_batchs(key, allrows) for b in bufs], axis=1)      # (rows, N, 4, H, W) float64
            assert arr.dtype == np.float64 and np.array_equal(arr, arr.astype(np.uint8))
            out[f"{tag}_ref_{key}"] = arr.astype(np.uint8)       # exact: the float64 values are the frame bytes
        for key in ("acts", "rewards", "terminals"):
            out[f"{tag}_ref_{key}"] = np.stack([b.encode_batchs(key, allrows) for b in bufs], axis=1)
        out[f"{tag}_frames"] = frames
        out[f"{tag}_acts"] = acts
        out[f"{tag}_args"] = np.array([N, rows, steps, H, W, horizon, max_frames, seed], dtype=np.int64)
        out[f"{tag}_top_size"] = np.array([bufs[0]._top, bufs[0]._size], dtype=np.int64)
        if tag == "single":                                      # the reference's own uniform sample (index stream + batch)
            np.random.seed(seed + 50)
            for k in range(3):
                batch = bufs[0].random_batch(7, ["obs", "next_obs", "acts", "rewards", "terminals"])
                for key, v in batch.items():
                    out[f"{tag}_batch{k}_{key}"] = v.astype(np.uint8) if key in ("obs", "next_obs") else v
    save("frame_dedup", **out)


def case_collect_offpolicy_norm():
    """VecCollector.train_one_epoch (collector/base.py:176-230) on a NormObs-wrapped synthetic env
    (env/base_wrapper.py:98-121 as get_vec_env applies it): the ring holds the NORMALISED observations env.step returns,
    statistics move every step, and after any reset the next policy input is the RAW array partial_reset returns
    (SURVEY Q14).  Followed by eval_one_epoch (:232-280) with the deep-copied normaliser in eval mode."""
    import gym
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.collector.base import VecCollector
    from torchrl.env.base_wrapper import NormObs
    from torchrl.replay_buffers.base import BaseReplayBuffer
    from oracle.synth_env import SynthVecEnvCPU
    out = {}
    for tag, N, steps, rows, horizon, max_frames, seed in (("env_limit", 8, 12, 16, 5, 999, 6), ("wrap", 4, 20, 7, 6, 5, 7)):
        D, A, H = 17, 6, 32
        torch.manual_seed(seed + 40)
        net = dict(hidden_shapes=[H, H], append_hidden_shapes=[], base_type=networks.MLPBase,
                   activation_func=torch.nn.ReLU)
        pf = policies.GuassianContPolicy(input_shape=D, output_shape=2 * A, tanh_action=True, **net)

        def mk():
            e = SynthVecEnvCPU(N, horizon=horizon)
            e.action_space = gym.spaces.Box(-1, 1, (A,))
            e.observation_space = gym.spaces.Box(-np.inf, np.inf, (D,))
            return NormObs(e)
        env, eval_env = mk(), mk()
        env.seed(seed)
        eval_env.seed(seed + 1)
        torch.manual_seed(seed)
        buf = BaseReplayBuffer(N * rows, env_nums=N)
        col = VecCollector(env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=torch.device("cpu"),
                           train_render=False, epoch_frames=N * steps, max_episode_frames=max_frames, eval_episodes=1)
        out.update(state_arrays(f"{tag}_pf_", pf))
        out[f"{tag}_ob0"] = np.asarray(col.current_ob).copy()
        noise_state = torch.get_rng_state()
        res = col.train_one_epoch()
        after = torch.get_rng_state()
        torch.set_rng_state(noise_state)
        out[f"{tag}_noise"] = torch.stack([torch.randn(N, A) for _ in range(steps)]).numpy()
        assert torch.equal(torch.get_rng_state(), after), "noise stream mismatch"
        for k in ("obs", "next_obs", "acts", "rewards", "terminals", "time_limits"):
            out[f"{tag}_buf_{k}"] = getattr(buf, "_" + k).copy()
        out[f"{tag}_top_size"] = np.array([buf._top, buf._size], dtype=np.int64)
        out[f"{tag}_train_epoch_reward"] = np.array(res["train_epoch_reward"])
        out[f"{tag}_train_rewards"] = np.array(res["train_rewards"], dtype=np.float64).reshape(-1)
        out[f"{tag}_current_ob"] = np.asarray(col.current_ob).copy()
        nz = env._obs_normalizer
        out[f"{tag}_state1"] = np.concatenate([nz._mean, nz._var, [nz._count]])
        ev = col.eval_one_epoch()
        out[f"{tag}_eval_rewards"] = np.array(ev["eval_rewards"], dtype=np.float64).reshape(-1)
        out[f"{tag}_eval_traj_length"] = np.array(ev["eval_traj_length"])
        assert np.array_equal(np.concatenate([nz._mean, nz._var, [nz._count]]), out[f"{tag}_state1"])   # eval does not update
        out[f"{tag}_args"] = np.array([N, steps, rows, horizon, max_frames, seed], dtype=np.int64)
    save("collect_offpolicy_norm", **out)


def _load_py_envs():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_py_envs", os.path.join(REPO, "torchrl_amd", "env", "py_envs.py"))
    py_envs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(py_envs)
    return py_envs


def case_collect_hostenv():
    """The reference's collectors over the reference's OWN VecEnv of single Python envs: what the ring really
    holds.  VecEnv.partial_reset writes the fresh observations into the array `step` just returned (env/vecenv.py:47-51)
    and both collectors add the sample AFTER the reset (collector/base.py:203-227, collector/on_policy.py:132-151): the
    stored `next_obs` row of every env that was reset in a step -- by `done` or by the collector's max_episode_frames,
    where `terminals` stays False -- is the RESET observation, not the one the env produced.  `*_reset_mask` logs the
    masks partial_reset was called with (row t, env i), `*_true_next_obs` what env.step returned."""
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.collector.base import VecCollector
    from torchrl.collector.on_policy import VecOnPolicyCollector
    from torchrl.env.vecenv import VecEnv
    from torchrl.replay_buffers.base import BaseReplayBuffer
    from torchrl.replay_buffers.on_policy import OnPolicyReplayBuffer
    from oracle.synth_env import SynthSingleEnvCPU
    import gym
    py_envs = _load_py_envs()

    class ShortPendulum(py_envs.PendulumEnv):
        """Episodes of 4 / 6 / 8 steps by seed, so that env time limits and the collector's limit interleave."""
        def seed(self, seed):
            super().seed(seed)
            self._max_episode_steps = 4 + 2 * (int(seed) % 3)
    short = ShortPendulum
    out = {}

    def make_env(kind, N, horizon, seed):
        if kind == "synth":
            count = iter(range(N))      # (the reference's list form trips over its own assert, vecenv.py:19: one factory)
            env = VecEnv(N, lambda: (lambda i: SynthSingleEnvCPU(seed * N + i, horizon + 2 * (i % 3)))(next(count)), ())
            env.envs[0].action_space = gym.spaces.Box(-1, 1, (6,))      # (`continuous` is an isinstance check on gym's Box)
            return env
        env = VecEnv(N, short if kind == "short_pendulum" else py_envs.PendulumEnv, ())
        env.seed(seed)
        return env

    def logged(env, steps_done):
        """Record every partial_reset mask and what env.step really returned."""
        masks, true_next = [], []
        reset, step = env.partial_reset, env.step

        def partial_reset(mask, **kw):
            masks.append(np.concatenate([[steps_done()], np.asarray(mask).reshape(-1).astype(np.int64)]))
            return reset(mask, **kw)

        def step_(actions):
            res = step(actions)
            true_next.append(np.array(res[0], copy=True))
            return res
        env.partial_reset, env.step = partial_reset, step_
        return masks, true_next

    # ---- off-policy (VecCollector, reparameterised tanh-Gaussian policy) ----
    for tag, kind, N, steps, rows, horizon, max_frames, seed in (
            ("off_pendulum_overlength", "pendulum", 4, 14, 20, 200, 5, 3),    # collector resets only: terminals all False
            ("off_pendulum_mixed", "short_pendulum", 4, 26, 30, 0, 5, 4),     # env time limits (4 / 6 / 8) and collector resets
            ("off_synth_wrap", "synth", 4, 20, 7, 4, 5, 5)):                  # 17/6 shape, horizons 4 / 6 / 8, the 7-row ring wraps twice
        env, eval_env = make_env(kind, N, horizon, seed), make_env(kind, N, horizon, seed + 1)
        D, A = env.observation_space.shape[0], env.action_space.shape[0]
        torch.manual_seed(seed + 40)
        net = dict(hidden_shapes=[32, 32], append_hidden_shapes=[], base_type=networks.MLPBase,
                   activation_func=torch.nn.ReLU)
        pf = policies.GuassianContPolicy(input_shape=D, output_shape=2 * A, tanh_action=True, **net)
        torch.manual_seed(seed)
        buf = BaseReplayBuffer(N * rows, env_nums=N)
        col = VecCollector(env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=torch.device("cpu"),
                           train_render=False, epoch_frames=N * steps, max_episode_frames=max_frames, eval_episodes=1)
        masks, true_next = logged(env, lambda: len(true_next) - 1)
        out.update(state_arrays(f"{tag}_pf_", pf))
        out[f"{tag}_ob0"] = np.asarray(col.current_ob).copy()
        res = col.train_one_epoch()
        for k in ("obs", "next_obs", "acts", "rewards", "terminals", "time_limits"):
            out[f"{tag}_buf_{k}"] = np.asarray(getattr(buf, "_" + k)).copy()
        out[f"{tag}_top_size"] = np.array([buf._top, buf._size], dtype=np.int64)
        out[f"{tag}_reset_mask"] = np.stack(masks)
        out[f"{tag}_true_next_obs"] = np.stack(true_next)
        out[f"{tag}_train_epoch_reward"] = np.array(res["train_epoch_reward"])
        out[f"{tag}_train_rewards"] = np.array(res["train_rewards"], dtype=np.float64).reshape(-1)
        out[f"{tag}_current_ob"] = np.asarray(col.current_ob).copy()
        out[f"{tag}_args"] = np.array([N, steps, rows, horizon, max_frames, seed], dtype=np.int64)
        # the deviation this fixture pins: stored next_obs == reset observation exactly on the reset rows
        flat_next, n_alias = np.stack(true_next), 0
        for m in masks:
            t, mask = int(m[0]), m[1:].astype(bool)
            if t >= steps - rows:                                     # row still in the ring
                stored = out[f"{tag}_buf_next_obs"][t % rows]
                assert np.array_equal(stored[~mask], flat_next[t][~mask])
                assert not np.array_equal(stored[mask], flat_next[t][mask])
                assert np.array_equal(stored[mask], out[f"{tag}_buf_obs"][(t + 1) % rows][mask]) or t == steps - 1
                n_alias += int(mask.sum())
        assert n_alias > 0

    # ---- off-policy, DISCRETE actions: epsilon-greedy Q policy (numpy global stream) on cart-poles that fall (`done` resets, terminals
    # True) and hit the collector's limit (terminals False) ----
    for tag, N, steps, rows, max_frames, seed in (("off_cartpole_dqn", 4, 40, 48, 12, 8),):
        env = VecEnv(N, py_envs.CartPoleEnv, ())
        eval_env = VecEnv(N, py_envs.CartPoleEnv, ())
        env.seed(seed)
        eval_env.seed(seed + 1)
        torch.manual_seed(seed + 60)
        qf = networks.Net(input_shape=4, output_shape=2, hidden_shapes=[32, 32], append_hidden_shapes=[],
                          base_type=networks.MLPBase, activation_func=torch.nn.ReLU)
        pf = policies.EpsilonGreedyDQNDiscretePolicy(qf, start_epsilon=0.8, end_epsilon=0.3, decay_frames=25, action_shape=2)
        np.random.seed(seed)
        buf = BaseReplayBuffer(N * rows, env_nums=N)
        col = VecCollector(env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=torch.device("cpu"),
                           train_render=False, epoch_frames=N * steps, max_episode_frames=max_frames, eval_episodes=1)
        masks, true_next = logged(env, lambda: len(true_next) - 1)
        out.update(state_arrays(f"{tag}_qf_", qf))
        out[f"{tag}_ob0"] = np.asarray(col.current_ob).copy()
        res = col.train_one_epoch()
        for k in ("obs", "next_obs", "acts", "rewards", "terminals", "time_limits"):
            out[f"{tag}_buf_{k}"] = np.asarray(getattr(buf, "_" + k)).copy()
        out[f"{tag}_top_size"] = np.array([buf._top, buf._size], dtype=np.int64)
        out[f"{tag}_reset_mask"] = np.stack(masks)
        out[f"{tag}_true_next_obs"] = np.stack(true_next)
        out[f"{tag}_train_epoch_reward"] = np.array(res["train_epoch_reward"])
        out[f"{tag}_train_rewards"] = np.array(res["train_rewards"], dtype=np.float64).reshape(-1)
        out[f"{tag}_current_ob"] = np.asarray(col.current_ob).copy()
        out[f"{tag}_epsilon_count"] = np.array([pf.epsilon, pf.count], dtype=np.float64)
        out[f"{tag}_args"] = np.array([N, steps, rows, max_frames, seed], dtype=np.int64)
        term = out[f"{tag}_buf_terminals"].reshape(rows, N)[:steps]
        reset = np.zeros((steps, N), dtype=bool)
        for m in masks:
            reset[int(m[0])] |= m[1:].astype(bool)
        assert term.any() and (reset & (term == 0)).any(), "want both fallen poles and over-length resets"
        for m in masks:
            t, mask = int(m[0]), m[1:].astype(bool)
            assert not np.array_equal(out[f"{tag}_buf_next_obs"][t][mask], np.stack(true_next)[t][mask])

    # ---- the reference's SubProcVecEnv (spawned workers; its partial_reset writes into the stacked array too,
    # env/subproc_vecenv.py:108-121) under VecCollector: env `done` resets and collector resets ----
    from torchrl.env.subproc_vecenv import SubProcVecEnv
    os.environ["PYTHONPATH"] = os.pathsep.join(p for p in sys.path if p)     # for the spawned workers
    for tag, N, procs, steps, rows, horizon, max_frames, seed in (("off_subproc_done", 6, 3, 11, 12, 3, 999, 9),
                                                                  ("off_subproc_overlength", 6, 3, 11, 12, 1000, 4, 10)):
        env = SubProcVecEnv(procs, N, SynthSingleEnvCPU, (0, horizon))      # (workers ignore `seed`, Q15: every env has seed 0)
        eval_env = None
        try:
            env.example_env.action_space = gym.spaces.Box(-1, 1, (6,))
            torch.manual_seed(seed + 40)
            net = dict(hidden_shapes=[32, 32], append_hidden_shapes=[], base_type=networks.MLPBase,
                       activation_func=torch.nn.ReLU)
            pf = policies.GuassianContPolicy(input_shape=17, output_shape=12, tanh_action=True, **net)
            torch.manual_seed(seed)
            buf = BaseReplayBuffer(N * rows, env_nums=N)
            col = VecCollector(env=env, eval_env=env, pf=pf, replay_buffer=buf, device=torch.device("cpu"),
                               train_render=False, epoch_frames=N * steps, max_episode_frames=max_frames, eval_episodes=1)
            masks, true_next = logged(env, lambda: len(true_next) - 1)
            out.update(state_arrays(f"{tag}_pf_", pf))
            out[f"{tag}_ob0"] = np.asarray(col.current_ob).copy()
            res = col.train_one_epoch()
        finally:
            env.close()
        for k in ("obs", "next_obs", "acts", "rewards", "terminals", "time_limits"):
            out[f"{tag}_buf_{k}"] = np.asarray(getattr(buf, "_" + k)).copy()
        out[f"{tag}_top_size"] = np.array([buf._top, buf._size], dtype=np.int64)
        out[f"{tag}_reset_mask"] = np.stack(masks)
        out[f"{tag}_true_next_obs"] = np.stack(true_next)
        out[f"{tag}_train_epoch_reward"] = np.array(res["train_epoch_reward"])
        out[f"{tag}_train_rewards"] = np.array(res["train_rewards"], dtype=np.float64).reshape(-1)
        out[f"{tag}_current_ob"] = np.asarray(col.current_ob).copy()
        out[f"{tag}_args"] = np.array([N, procs, steps, rows, horizon, max_frames, seed], dtype=np.int64)
        for m in masks:
            t, mask = int(m[0]), m[1:].astype(bool)
            assert not np.array_equal(out[f"{tag}_buf_next_obs"][t][mask], np.stack(true_next)[t][mask])

    # ---- on-policy (VecOnPolicyCollector): the bootstrap value uses the TRUE next observation, the row the reset one ----
    for tag, kind, N, T, horizon, max_frames, seed in (
            ("on_pendulum_mixed", "short_pendulum", 4, 24, 0, 5, 6),
            ("on_synth_mixed", "synth", 8, 16, 4, 5, 7)):
        env, eval_env = make_env(kind, N, horizon, seed), make_env(kind, N, horizon, seed + 1)
        D, A = env.observation_space.shape[0], env.action_space.shape[0]
        pf, vf = build_nets(D, A, 64, seed=seed + 20)
        torch.manual_seed(seed)
        buf = OnPolicyReplayBuffer(N * T, env_nums=N, time_limit_filter=True)
        col = VecOnPolicyCollector(vf, env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=torch.device("cpu"),
                                   train_render=False, epoch_frames=N * T, max_episode_frames=max_frames, eval_episodes=1)
        masks, true_next = logged(env, lambda: len(true_next) - 1)
        out.update(state_arrays(f"{tag}_pf_", pf))
        out.update(state_arrays(f"{tag}_vf_", vf))
        out[f"{tag}_ob0"] = np.asarray(col.current_ob).copy()
        res = col.train_one_epoch()
        for k in ("obs", "next_obs", "acts", "values", "rewards", "terminals", "time_limits"):
            out[f"{tag}_buf_{k}"] = np.asarray(getattr(buf, "_" + k)).copy()
        out[f"{tag}_reset_mask"] = np.stack(masks)
        out[f"{tag}_true_next_obs"] = np.stack(true_next)
        out[f"{tag}_train_epoch_reward"] = np.array(res["train_epoch_reward"])
        out[f"{tag}_train_rewards"] = np.array(res["train_rewards"], dtype=np.float64).reshape(-1)
        out[f"{tag}_current_ob"] = np.asarray(col.current_ob).copy()
        out[f"{tag}_args"] = np.array([N, T, horizon, max_frames, seed], dtype=np.int64)
        for m in masks:
            t, mask = int(m[0]), m[1:].astype(bool)
            assert not np.array_equal(out[f"{tag}_buf_next_obs"][t][mask], np.stack(true_next)[t][mask])
    save("collect_hostenv", **out)


CNN_INIT_CONVS = [[16, [8, 8], [4, 4], [0, 0]], [32, [4, 4], [2, 2], [0, 0]], [64, [3, 3], [1, 1], [0, 0]]]


def case_cnn_init():
    """networks.Net over CNNBase (networks/base.py:59-107, nets.py) and the orthogonal initialiser (init.py:40-47) under
    torch.manual_seed: the cfg 5 trunk (conv 16/32/64) on a 4 x 36 x 36 input (64 features) + fc 512 + a 6-wide head; a
    QNet-style MLP with orthogonal_init; one forward pass of the conv net on a fixed uint8-valued input."""
    import torchrl.networks as networks
    out = {}
    torch.manual_seed(43)
    qf = networks.Net(output_shape=6, base_type=networks.CNNBase, append_hidden_shapes=[512],
                      activation_func=torch.nn.ReLU, input_shape=(4, 36, 36), hidden_shapes=CNN_INIT_CONVS)
    out.update(state_arrays("cnn_", qf))
    x = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (3, 4, 36, 36)).astype(np.float32))
    out["input_x"] = x.numpy()
    out["output_y"] = qf(x).detach().numpy()
    torch.manual_seed(44)
    mlp = networks.Net(input_shape=(11,), output_shape=3, hidden_shapes=[32, 32], append_hidden_shapes=[],
                       base_type=networks.MLPBase, activation_func=torch.nn.ReLU,
                       init_func=networks.orthogonal_init, net_last_init_func=networks.orthogonal_init)
    out.update(state_arrays("ortho_", mlp))
    save("cnn_init", **out)


CASES = {"cnn_init": case_cnn_init, "collect_hostenv": case_collect_hostenv, "collect_offpolicy_norm": case_collect_offpolicy_norm, "frame_dedup": case_frame_dedup, "collect_offpolicy": case_collect_offpolicy, "subproc_vecenv": case_subproc_vecenv, "eps_greedy": case_eps_greedy, "eval_epoch": case_eval_epoch, "vecenv": case_vecenv, "gae": case_gae, "index_streams": case_index_streams, "init": case_init, "ppo_update": case_ppo_update,
         "collect_epoch": case_collect_and_epoch, "twin_sac_q": case_twin_sac_q, "dqn": case_dqn,
         "obs_norm": case_obs_norm, "a2c_update": case_a2c_update, "ddpg_td3": case_ddpg_td3, "vmpo_update": case_vmpo_update, "trpo_update": case_trpo_update}

def check(names):
    """Regenerate into a scratch directory and compare with the committed fixtures (bit for bit)."""
    global HERE
    committed, HERE = HERE, tempfile.mkdtemp(prefix="trl_golden_check_")
    bad = 0
    for name in names:
        CASES[name]()
    for fname in sorted(f for f in os.listdir(HERE) if f.endswith(".npz")):
        new, old = np.load(os.path.join(HERE, fname)), np.load(os.path.join(committed, fname))
        same = sorted(new.files) == sorted(old.files) and all(
            new[k].shape == old[k].shape and np.array_equal(new[k], old[k], equal_nan=new[k].dtype.kind == "f")
            for k in new.files if k != "meta")
        bad += not same
        print("%-24s %s" % (fname, "identical" if same else "DIFFERS"))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    install_stubs()
    argv = [a for a in sys.argv[1:] if a != "--check"]
    if "--check" in sys.argv[1:]:                            # python make_golden.py --check [case ...]
        check(argv or list(CASES))
    for name in (argv or list(CASES)):                       # python make_golden.py [case ...]
        CASES[name]()
