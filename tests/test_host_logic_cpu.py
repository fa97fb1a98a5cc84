"""Host-side logic of the product classes that needs no GPU: index streams, ring bookkeeping,
argument checks, API surface / alias package, config + logger.  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_alias_package_resolves_to_same_modules():
    import torchrl
    import torchrl_amd
    import torchrl.algo
    import torchrl_amd.algo
    from torchrl.collector.on_policy import VecOnPolicyCollector
    from torchrl_amd.collector.on_policy import VecOnPolicyCollector as V2
    assert torchrl is torchrl_amd and torchrl.algo is torchrl_amd.algo and VecOnPolicyCollector is V2
    # the import lines of the reference's examples/ppo_continuous_vec.py:8-18
    from torchrl.utils import get_args, get_params, Logger                  # noqa: F401
    from torchrl.replay_buffers.on_policy import OnPolicyReplayBuffer      # noqa: F401
    import torchrl.policies as policies
    import torchrl.networks as networks
    from torchrl.algo import PPO                                           # noqa: F401
    from torchrl.env import get_vec_env                                    # noqa: F401
    import gym
    assert hasattr(gym.spaces, "Box")
    assert hasattr(policies, "GuassianContPolicyBasicBias") and hasattr(networks, "MLPBase")


def test_epoch_row_indices_is_the_reference_permutation_stream(golden):
    """Same numpy calls in the same order as one_iteration (on_policy.py:76-78): E permutations."""
    from torchrl.replay_buffers.on_policy import OnPolicyReplayBuffer
    g = golden("index_streams")
    T, N, B, E, seed = (int(x) for x in g["oi_args"])
    buf = OnPolicyReplayBuffer(T * N, env_nums=N)
    np.random.seed(seed)
    got = np.concatenate([buf.epoch_row_indices(B, True) for _ in range(E)])
    np.random.seed(seed)
    want = np.concatenate([np.random.permutation(T).reshape(-1, B // N) for _ in range(E)])
    assert np.array_equal(got, want) and got.dtype == np.int64
    assert np.array_equal(buf.epoch_row_indices(B, False).reshape(-1), np.arange(T))
    # data check against the reference's gathered batches
    cat = np.concatenate([g["oi_obs"], g["oi_acts"], g["oi_advs"]], -1)
    for k, rows in enumerate(got):
        assert np.array_equal(cat[rows].reshape(B, -1), g["oi_batches"][k])
    with pytest.raises(AssertionError, match="dividable"):
        buf.epoch_row_indices(B + 1, True)
    with pytest.raises(ValueError, match="multiple"):
        OnPolicyReplayBuffer(10 * N, env_nums=N).epoch_row_indices(4 * N, True)


def test_ring_bookkeeping_matches_reference(golden):
    from torchrl.replay_buffers import BaseReplayBuffer
    g = golden("index_streams")
    size, N, B, seed = (int(x) for x in g["ring_args"])
    ring = BaseReplayBuffer(size, env_nums=N)
    assert ring._max_replay_buffer_size == size // N
    for t in range(7):
        ring._advance()
        assert ring._size == g["ring_sizes"][t] and ring._top == g["ring_tops"][t]
    ring._advance(3)
    assert ring._top == (g["ring_tops"][6] + 3) % (size // N) and ring.num_steps_can_sample() == size // N


def test_net_structure_names_and_mlp2_spec():
    import torchrl.networks as networks
    import torchrl.policies as policies
    net = dict(hidden_shapes=[64, 64], append_hidden_shapes=[], base_type=networks.MLPBase,
               activation_func=torch.nn.Tanh)
    pf = policies.GuassianContPolicyBasicBias(input_shape=17, output_shape=6, tanh_action=True, **net)
    vf = networks.Net(input_shape=(17,), output_shape=1, **net)
    assert sorted(pf.state_dict()) == ["base.seq_fcs.0.bias", "base.seq_fcs.0.weight", "base.seq_fcs.2.bias",
                                       "base.seq_fcs.2.weight", "logstd", "seq_append_fcs.0.bias",
                                       "seq_append_fcs.0.weight"]
    assert pf.mlp2_spec() == (17, 64, 6, 0) and vf.mlp2_spec() == (17, 64, 1, 0)
    assert sum(p.numel() for p in pf.parameters()) == 5708 and sum(p.numel() for p in vf.parameters()) == 5377
    assert np.allclose(pf.logstd.detach().numpy(), np.log(0.125))
    # reference init conventions (networks/init.py): hidden U(+-1/8), bias 0.1, head U(+-3e-3)
    sd = vf.state_dict()
    assert sd["base.seq_fcs.0.weight"].abs().max() <= 0.125 and torch.all(sd["base.seq_fcs.2.bias"] == 0.1)
    assert sd["seq_append_fcs.0.weight"].abs().max() <= 3e-3
    # flat view aliases the parameters
    flat = vf.flat_params()
    assert flat.numel() == 5377 and flat.data_ptr() == vf.base.seq_fcs[0].weight.data_ptr()
    with torch.no_grad():
        flat[:] = 0.5
    assert float(vf.seq_append_fcs[0].bias) == 0.5
    odd = networks.Net(input_shape=(17,), output_shape=1, hidden_shapes=[400, 300], append_hidden_shapes=[],
                       base_type=networks.MLPBase)
    assert odd.mlp2_spec() is None
    # CPU / autograd forward is the plain module graph
    out = odd(torch.zeros(3, 17))
    assert out.shape == (3, 1) and out.requires_grad


def test_product_path_refuses_cpu_and_unknown_envs():
    from torchrl_amd import _C
    from torchrl.env import get_vec_env
    with pytest.raises(ValueError, match="unknown env id"):
        get_vec_env("HalfCheetah-v2", {"reward_scale": 1, "obs_norm": False}, 4)
    with pytest.raises(NotImplementedError, match="rew_norm"):
        get_vec_env("SynthHalfCheetah-v0", {"reward_scale": 1, "obs_norm": False, "rew_norm": {}}, 4)
    from torchrl.replay_buffers.on_policy import OnPolicyReplayBuffer
    buf = OnPolicyReplayBuffer(8, env_nums=2, device="cpu")
    for k in ("rewards", "values", "terminals", "time_limits"):
        buf._ensure_key(k, (2, 1))
    with pytest.raises(_C.TrlError, match="no CPU path"):
        buf.generalized_advantage_estimation(torch.zeros(2, 1), 0.99, 0.95)


def test_args_params_logger(tmp_path):
    from torchrl.utils import get_params, Logger
    from torchrl.utils.args import get_args
    args = get_args(["--config", "x.json", "--vec_env_nums", "2048", "--seed", "3", "--no_cuda"])
    assert args.vec_env_nums == 2048 and args.seed == 3 and args.cuda is False
    params = get_params(os.path.join(REPO, "config", "ppo_synth_halfcheetah.json"))
    assert params["replay_buffer"]["size"] == 2048 * 128 and params["general_setting"]["batch_size"] == 65536
    log = Logger("exp", "SynthHalfCheetah-v0", 0, dict(params), str(tmp_path), overwrite=True)
    for v in (1.0, 3.0):
        log.add_update_info({"Training/vf_loss": v})
    log.add_epoch_info(0, 100, 1.5, {"Train_Epoch_Reward": 2.0})
    log.add_epoch_info(1, 200, 1.5, {"Train_Epoch_Reward": 4.0})
    log.finish()
    rows = open(os.path.join(log.work_dir, "log.csv")).read().strip().split("\n")
    assert rows[0].startswith("EPOCH,Time Consumed,Total Frames,Train_Epoch_Reward,Training/vf_loss_Mean")
    assert rows[1].split(",")[4] == "2.00000" and len(rows) == 3       # '{:.5f}', as the reference writes its csv values
    assert json.load(open(os.path.join(log.work_dir, "params.json")))["env_name"] == "SynthHalfCheetah-v0"
    with pytest.raises(AssertionError, match="overwrite"):
        Logger("exp", "SynthHalfCheetah-v0", 0, dict(params), str(tmp_path), overwrite=False)


def test_statistics_ring_and_index_slab_bookkeeping():
    """algo/off_policy/_deferred.py on CPU tensors: update u sits in row u % slots; when more updates are launched than
    unread rows fit, the pending handles are re-pointed at a copy; reads come back in handle order; the index slab is
    {first update count, sets, idx...}."""
    from torchrl_amd.algo.off_policy._deferred import IndexSlab, StatRing
    ring = StatRing(4, 2, torch.float64, "cpu")

    def launch(first, count):                                            # what an engine does around `count` updates
        ring.make_room(count)
        for u in range(first, first + count):
            ring.t[u % 4] = torch.tensor([float(u), -float(u)])          # (the device writes the row; here the host)
        return ring.handles(first, count)
    h = launch(0, 3)
    assert [r for _, r in h] == [0, 1, 2] and ring.read(h).tolist() == [[0, 0], [1, -1], [2, -2]]
    h1 = launch(3, 3)                                                    # rows 3, 0, 1: everything before was read
    assert h1[0][0] is h[0][0] and [r for _, r in h1] == [3, 0, 1]
    h2 = launch(6, 2)                                                    # 3 unread + 2 > 4 rows: h1 moves to a copy
    assert h2[0][0] is not h1[0][0] and h1[0][0].t is not ring.t and h2[0][0].t is ring.t
    mixed = [h2[1], h1[0], h1[2], h2[0]]
    assert ring.read(mixed).tolist() == [[7, -7], [3, -3], [5, -5], [6, -6]]
    assert ring.read(h2).tolist() == [[6, -6], [7, -7]] and ring.read([]).shape == (0, 2)
    h3 = launch(8, 4)                                                    # h2 was read in full: no copy needed
    assert h3[0][0] is h2[0][0] and ring.read(h3)[:, 0].tolist() == [8, 9, 10, 11]
    # handles resolved piecemeal (every epoch's on their own) and twice: rows are free once read, whatever the order
    h4 = launch(12, 2)                                                   # rows 0, 1
    assert ring.read(h4[:1]).tolist() == [[12, -12]] and ring.read(h4[:1]).tolist() == [[12, -12]]
    h5 = launch(14, 3)                                                   # rows 2, 3, 0: row 0 was read, row 1 is not touched
    assert h5[0][0] is h4[0][0]
    h6 = launch(17, 4)                                                   # rows 1, 2, 3, 0: unread updates 13, 14, 15, 16
    assert h6[0][0] is not h5[0][0] and ring.read([h4[1]] + h5)[:, 0].tolist() == [13, 14, 15, 16]
    assert ring.read(h6)[:, 0].tolist() == [17, 18, 19, 20]

    slab = IndexSlab("cpu")
    idx = np.arange(12).reshape(3, 4)
    dev = slab.upload(17, idx)
    assert dev.dtype == torch.int64 and dev.tolist() == [17, 3] + list(range(12))
    assert slab.upload(20, idx + 1) is dev and dev[:3].tolist() == [20, 3, 1]


def test_logger_takes_deferred_update_infos_in_order(tmp_path):
    """Logger.add_update_infos_later: dicts of launched-but-not-awaited updates are taken when the next dict arrives or the
    next row is written, in arrival order; the row equals the one the immediate calls give."""
    from torchrl_amd.utils import Logger
    rows = []
    for deferred in (False, True):
        log = Logger("exp%d" % deferred, "Env-v0", 0, {}, log_dir=str(tmp_path), overwrite=True)
        a, b, c = [{"x": 1.0, "y": 2.0}, {"x": 3.0, "y": 0.0}], [{"x": 5.0, "y": 1.0}], {"x": -1.0, "y": 4.0}
        calls = []
        if deferred:
            log.add_update_infos_later(lambda: calls.append("a") or a)
            log.add_update_infos_later(lambda: calls.append("b") or b)
            assert calls == [] and log.update_count == 0
            log.add_update_info(c)
            assert calls == ["a", "b"] and log.stored_infos["x"] == [1.0, 3.0, 5.0, -1.0]
            log.add_update_infos_later(lambda: calls.append("a2") or a)
        else:
            for d in a + b + [c] + a:
                log.add_update_info(d)
        log.add_epoch_info(0, 10, 1.0, {"r": 1.0})
        assert log.update_count == 6 and log.stored_infos == {}
        rows.append(open(log.csv_file_path).read().splitlines())
        log.finish()
    assert rows[0][0] == rows[1][0] and rows[0][1].split(",")[3:] == rows[1][1].split(",")[3:]


def test_linear_lr_schedule_and_param_copy():
    from torchrl.algo import utils as atu
    lin = torch.nn.Linear(3, 2)
    opt = torch.optim.Adam(lin.parameters(), lr=3e-4)
    atu.update_linear_schedule(opt, 3, 10, 3e-4)
    assert abs(opt.param_groups[0]["lr"] - 3e-4 * 0.7) < 1e-12
    tgt = torch.nn.Linear(3, 2)
    atu.copy_model_params_from_to(lin, tgt)
    assert torch.equal(tgt.weight, lin.weight)
    atu.soft_update_from_to(torch.nn.Linear(3, 2), tgt, 0.0)
    assert torch.equal(tgt.weight, lin.weight)


def test_vecenv_protocol_and_pendulum_env():
    """torchrl.env.VecEnv over single Python envs (reference protocol, vecenv.py:6-78) -- pure host code."""
    import numpy as np
    from oracle.synth_env import SynthSingleEnvCPU
    from torchrl_amd.env.py_envs import PendulumEnv
    from torchrl_amd.env.vecenv import VecEnv
    N, seed = 5, 2
    env = VecEnv(N, [SynthSingleEnvCPU] * N, [(seed * N + i, 3) for i in range(N)])
    obs = env.reset()
    assert obs.shape == (5, 17) and env.env_nums == 5 and env.observation_space.shape == (17,)
    nxt, rew, done, infos = env.step(np.zeros((5, 6)))
    assert nxt.shape == (5, 17) and rew.shape == (5, 1) and done.shape == (5, 1) and done.dtype == bool
    assert infos["time_limit"].shape == (5,)
    kept = nxt.copy()
    whole = env.partial_reset(np.array([0, 1, 0, 0, 1], dtype=bool))   # the WHOLE array comes back (vecenv.py:47-51)
    assert whole.shape == (5, 17) and np.array_equal(whole[0], kept[0]) and not np.array_equal(whole[1], kept[1])
    # ... and, as in the reference (vecenv.py:50 assigns into self._obs), it IS the array step() returned: a collector that
    # stores `next_obs` after the reset stores the reset observations (tests/golden/collect_hostenv.npz)
    assert VecEnv.alias_reset_obs and whole is nxt and not np.array_equal(nxt, kept) and np.array_equal(nxt[[0, 2, 3]], kept[[0, 2, 3]])
    env.alias_reset_obs = False                                        # opt-out: what step() returned stays what the env produced
    nxt, _, _, _ = env.step(np.zeros((5, 6)))
    kept = nxt.copy()
    whole = env.partial_reset(np.array([1, 0, 0, 0, 0], dtype=bool))
    assert whole is not nxt and np.array_equal(nxt, kept) and not np.array_equal(whole[0], kept[0])
    assert env.horizon == 3                                            # unknown attributes fall through to envs[0]
    with __import__("pytest").raises(ValueError):
        VecEnv(3, [SynthSingleEnvCPU] * 2, [(0, 3)] * 2)
    pend = VecEnv(4, PendulumEnv, ())
    pend.seed(7)
    o = pend.reset()
    assert o.shape == (4, 3) and np.allclose(o[:, 0] ** 2 + o[:, 1] ** 2, 1.0, atol=1e-6)
    for _ in range(200):
        o, r, d, info = pend.step(np.ones((4, 1)) * 0.5)
    assert d.all() and info["time_limit"].all() and (r <= 0).all()     # 200-step time limit, costs only


def test_vecenv_reproduces_the_reference_vecenv(golden):
    """The reference's VecEnv, run over the same single envs through the same call sequence (tests/golden/vecenv.npz):
    every observation, reward, done flag, time-limit info and partial_reset result, bit for bit -- for the in-process
    VecEnv and for the spawned-worker SubProcVecEnv."""
    import importlib.util
    import numpy as np
    from torchrl_amd.env import SubProcVecEnv, VecEnv
    from torchrl_amd.env.py_envs import CartPoleEnv, PendulumEnv
    spec = importlib.util.spec_from_file_location("_make_golden", os.path.join(REPO, "tests", "golden", "make_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    g = golden("vecenv")
    for kind, cls in (("pendulum", PendulumEnv), ("cartpole", CartPoleEnv)):
        N, steps = (int(x) for x in g[f"{kind}_args"])
        for make in (lambda: VecEnv(N, cls, ()), lambda: SubProcVecEnv(2, N, cls, ())):
            env = make()
            try:
                rec = gen.vecenv_script(env, kind, N, steps)
            finally:
                env.close()
            assert rec["mask"].shape[0] >= 5 and rec["done"].any() and (kind != "pendulum" or rec["tl"].any())
            for k, v in rec.items():
                assert v.dtype == g[f"{kind}_{k}"].dtype and np.array_equal(v, g[f"{kind}_{k}"]), (kind, k)


def test_subproc_vecenv_matches_vecenv():
    """SubProcVecEnv (spawned workers, pipes) is the same function of (seed, actions) as the in-process VecEnv."""
    import numpy as np
    from torchrl_amd.env import SubProcVecEnv, VecEnv
    from torchrl_amd.env.py_envs import CartPoleEnv, PendulumEnv
    for cls, act in ((PendulumEnv, lambda rs: rs.uniform(-1, 1, size=(4, 1))), (CartPoleEnv, lambda rs: rs.randint(0, 2, size=(4,)))):
        a, b = VecEnv(4, cls, ()), SubProcVecEnv(2, 4, cls, ())
        try:
            a.seed(11); b.seed(11)
            a.train(); b.train()
            assert np.array_equal(a.reset(), b.reset())
            assert b.observation_space.shape == a.observation_space.shape
            rs = np.random.RandomState(0)
            for t in range(40):
                acts = act(rs)
                (oa, ra, da, ia), (ob, rb, db, ib) = a.step(acts), b.step(acts)
                assert np.array_equal(oa, ob) and np.array_equal(ra, rb) and np.array_equal(da, db)
                assert sorted(ia) == sorted(ib) and all(np.array_equal(ia[k], ib[k]) for k in ia)
                if da.any() or t % 7 == 3:
                    mask = da.reshape(-1) | (np.arange(4) == t % 4)
                    assert np.array_equal(a.partial_reset(mask), b.partial_reset(mask))
        finally:
            b.close()
    import pytest
    with pytest.raises(ValueError):
        SubProcVecEnv(3, 4, PendulumEnv, ())


def test_reference_import_surface():
    """Every name SURVEY.md 8(b) lists as imported by the reference's example scripts resolves through the `torchrl`
    alias; the ones without a kernel path fail loudly at construction, not at import."""
    import importlib
    import pytest
    from torchrl_amd import _C
    names = {"torchrl.utils": ["get_args", "get_params", "Logger"],
             "torchrl.env": ["get_vec_env", "get_env", "get_subprocvec_env", "VecEnv", "SubProcVecEnv"],
             "torchrl.replay_buffers": ["BaseReplayBuffer"], "torchrl.replay_buffers.on_policy": ["OnPolicyReplayBuffer"],
             "torchrl.collector": ["VecCollector", "BaseCollector"], "torchrl.collector.base": ["VecCollector"],
             "torchrl.collector.on_policy": ["VecOnPolicyCollector", "OnPolicyCollectorBase"],
             "torchrl.algo": ["PPO", "A2C", "TwinSACQ", "DQN", "DDPG", "TD3", "TRPO", "VMPO"],
             "torchrl.networks": ["MLPBase", "CNNBase", "Net", "QNet", "init"],
             "torchrl.policies": ["GuassianContPolicyBasicBias", "GuassianContPolicy", "FixGuassianContPolicy",
                                  "EpsilonGreedyDQNDiscretePolicy", "CategoricalDisPolicy"]}
    for mod, attrs in names.items():
        m = importlib.import_module(mod)
        for a in attrs:
            assert hasattr(m, a), (mod, a)
    from torchrl.algo import Reinforce
    from torchrl.policies import CategoricalDisPolicy
    for cls in (Reinforce, CategoricalDisPolicy):
        with pytest.raises(_C.TrlError, match="not built"):
            cls()


def test_oracle_is_test_infrastructure_only():
    """oracle/ may be imported by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg -- never by the
    package or by any other script: scan every Python source for an import of it."""
    import ast
    allowed = {"bench.py", "__graft_entry__.py"}
    offenders = []
    for root, dirs, files in os.walk(REPO):
        dirs[:] = [d for d in dirs if d not in (".git", "__pycache__", "gpurun_out", "tests", "oracle", ".pytest_cache")]
        for name in files:
            if not name.endswith(".py"):
                continue
            path = os.path.join(root, name)
            rel = os.path.relpath(path, REPO)
            tree = ast.parse(open(path).read())
            for node in ast.walk(tree):
                mods = [a.name for a in node.names] if isinstance(node, ast.Import) else \
                    [node.module or ""] if isinstance(node, ast.ImportFrom) and node.level == 0 else []
                if any(m == "oracle" or m.startswith("oracle.") for m in mods) and rel not in allowed:
                    offenders.append(rel)
    assert not offenders, offenders
    # and inside the two allowed files, only in the functions that are the checker legs
    for rel, legs in (("bench.py", {"cpu_baseline", "cpu_baseline_full", "cpu_baseline_offpolicy"}), ("__graft_entry__.py", {"smoke", "build"})):
        tree = ast.parse(open(os.path.join(REPO, rel)).read())
        for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
            uses = any(isinstance(n, (ast.Import, ast.ImportFrom)) and
                       any((getattr(n, "module", None) or a.name).split(".")[0] == "oracle" for a in n.names)
                       for n in ast.walk(fn))
            assert not uses or fn.name in legs, (rel, fn.name)


def test_epoch_driver_protocol(tmp_path):
    """RLAlgo.train (rl_algo.py:96-167) with stand-in collector / logger: call order, evaluation and snapshot cadence,
    `best` snapshots, the logger row's keys, timers restarting after every logged row, file names on disk."""
    import torchrl_amd  # noqa: F401  (registers the stand-in `gym` when none is installed)
    import gym
    from torchrl_amd.algo.rl_algo import RLAlgo
    calls, rows = [], []

    class Collector:
        epoch_frames = 40
        def train_one_epoch(self):
            calls.append("collect")
            return {"train_rewards": [1.0, 3.0] if len([c for c in calls if c == "collect"]) > 1 else [],
                    "train_epoch_reward": 7.5}
        def eval_one_epoch(self):
            calls.append("eval")
            n = len([c for c in calls if c == "eval"])
            return {"eval_rewards": [float(n), float(n) + 1.0] if n != 2 else [-5.0], "eval_traj_length": 9.0}
        def terminate(self): calls.append("terminate")

    class Log:
        def add_epoch_info(self, epoch, frames, dt, infos, csv_write=True): rows.append((epoch, frames, dict(infos)))
        def add_update_info(self, d): pass
        def log(self, *a): pass
        def finish(self): calls.append("finish")

    class Env:
        action_space = gym.spaces.Box(-1, 1, (2,))
        _obs_normalizer = {"mean": 0.5}

    class Algo(RLAlgo):
        def __init__(self, **kw):
            super().__init__(**kw)
            self.net = torch.nn.Linear(2, 2)
        snapshot_networks = property(lambda self: [("pf", self.net)])
        def update_per_epoch(self): calls.append("update")
        def finish_epoch(self): return {"extra": 1}

    save = tmp_path / "model"
    algo = Algo(env=Env(), replay_buffer=None, collector=Collector(), logger=Log(), num_epochs=5, eval_interval=2,
                save_interval=3, save_dir=str(save), device="cpu")
    assert algo.continuous and algo.epoch_frames == 40
    algo.train()
    assert calls == ["collect", "update", "eval", "collect", "update", "collect", "update", "eval",
                     "collect", "update", "collect", "update", "eval", "terminate", "finish"]
    assert [(e, f) for e, f, _ in rows] == [(0, 40), (2, 120), (4, 200)]
    first = rows[0][2]
    assert list(first)[:6] == ["Running_Average_Rewards", "Train_Epoch_Reward", "Running_Training_Average_Rewards",
                               "Explore_Time", "Train___Time", "Eval____Time"]
    assert first["eval_traj_length"] == 9.0 and first["extra"] == 1 and "eval_rewards" not in first
    assert np.isnan(first["Running_Training_Average_Rewards"]) and rows[1][2]["Running_Training_Average_Rewards"] == 2.0
    assert first["Running_Average_Rewards"] == 1.5 and rows[1][2]["Running_Average_Rewards"] == (1 + 2 - 5) / 3
    assert algo.best_eval == 3.5 and algo.explore_time == 0 and algo.train_time == 0      # evals: 1.5, -5.0, 3.5
    names = sorted(os.listdir(save))
    assert names == sorted(["model_pf_best.pth", "model_pf_0.pth", "model_pf_3.pth", "model_pf_finish.pth",
                            "_obs_normalizer_best.pkl", "_obs_normalizer_0.pkl", "_obs_normalizer_3.pkl",
                            "_obs_normalizer_finish.pkl"])
    assert sorted(torch.load(save / "model_pf_finish.pth")) == ["bias", "weight"]


def test_off_policy_epoch_driver_protocol():
    """OffRLAlgo (off_rl_algo.py:8-84) with stand-ins: opt_times x {sample -> update -> log}, the fixed-address batch
    handed to the replay gather when the engine has one, collection-only pretrain epochs and their frame count."""
    import torchrl_amd  # noqa: F401
    import gym
    from torchrl_amd.algo.off_policy.off_rl_algo import OffRLAlgo
    events, rows, logged = [], [], []

    class Replay:
        def random_batch(self, batch_size, keys, out=None):
            events.append(("sample", batch_size, tuple(keys), out))
            return {"n": len(events)}
        def num_steps_can_sample(self): return 100

    class Collector:
        epoch_frames = 16
        def train_one_epoch(self):
            events.append("collect")
            return {"train_rewards": [2.0], "train_epoch_reward": 1.0}

    class Log:
        def add_epoch_info(self, epoch, frames, dt, infos, csv_write=True): rows.append((epoch, frames, dict(infos), csv_write))
        def add_update_info(self, d): logged.append(d)
        def log(self, msg): events.append(msg)

    class Env:
        action_space = gym.spaces.Discrete(3)

    class Algo(OffRLAlgo):
        def update(self, batch): return {"seen": batch["n"]}

    algo = Algo(env=Env(), replay_buffer=Replay(), collector=Collector(), logger=Log(), batch_size=32, opt_times=3,
                pretrain_epochs=2, min_pool=10, device="cpu")
    assert not algo.continuous and algo.sample_key == ["obs", "next_obs", "acts", "rewards", "terminals"]
    algo.pretrain()
    assert events == ["collect", "collect", "Finished Pretrain"] and algo.pretrain_frames == 32
    assert [(e, f, c) for e, f, _, c in rows] == [(0, 16, False), (1, 32, False)]
    assert rows[1][2] == {"Train_Epoch_Reward": 1.0, "Running_Training_Average_Rewards": 2.0}
    del events[:]
    algo.update_per_epoch()
    assert events == [("sample", 32, tuple(algo.sample_key), None)] * 3 and [d["seen"] for d in logged] == [1, 2, 3]
    fixed = {"obs": object()}
    algo.static_batch = lambda: fixed                              # an engine with fixed-address inputs
    algo.update_per_timestep()                                     # 100 > max(min_pool, batch_size): opt_times updates
    assert [e[3] for e in events[3:]] == [fixed] * 3
    algo.min_pool = 1000
    algo.update_per_timestep()
    assert len(events) == 6


def test_whole_epoch_hook_of_the_off_policy_driver_and_its_index_draw():
    """`update_epoch_deferred(opt_times)`: taken when the engine accepts (no per-update sample), else the per-update loop;
    `draw_indices` = the index sets `count` random_batch calls would draw, generator state included."""
    import torchrl_amd  # noqa: F401
    import gym
    from torchrl_amd.algo.off_policy.off_rl_algo import OffRLAlgo
    from torchrl_amd.replay_buffers import BaseReplayBuffer
    calls, logged = [], []

    class Replay:
        def random_batch(self, batch_size, keys, out=None):
            calls.append("sample")
            return {}
        def num_steps_can_sample(self): return 100

    class Log:
        def add_update_info(self, d): logged.append(d)

    class Env:
        action_space = gym.spaces.Discrete(3)

    class Collector:
        epoch_frames = 16

    class Algo(OffRLAlgo):
        accept = True
        def update_deferred(self, batch): calls.append("one"); return len(calls)
        def update_epoch_deferred(self, count): calls.append(("epoch", count)); return list(range(count)) if self.accept else None
        def resolve_updates(self, handles): return [{"h": h} for h in handles]

    algo = Algo(env=Env(), replay_buffer=Replay(), collector=Collector(), logger=Log(), batch_size=32, opt_times=3,
                device="cpu")
    algo.update_per_epoch()
    assert calls == [("epoch", 3)] and logged == [{"h": 0}, {"h": 1}, {"h": 2}]
    del calls[:], logged[:]
    algo.accept = False
    algo.update_per_epoch()
    assert calls == [("epoch", 3), "sample", "one", "sample", "one", "sample", "one"] and len(logged) == 3

    buf = BaseReplayBuffer(40 * 8, env_nums=8, device="cpu")
    buf._size = 37
    np.random.seed(11)
    want = np.stack([np.random.randint(0, 37, 4) for _ in range(5)])
    after = np.random.randint(0, 1 << 30)
    np.random.seed(11)
    got = buf.draw_indices(32, 5)
    assert got.dtype == np.int64 and (got == want).all() and np.random.randint(0, 1 << 30) == after


def test_one_block_of_host_noise_equals_per_step_draws():
    """VecOnPolicyCollector._host_noise draws the whole rollout's exploration noise in ONE torch.randn call when
    N * A is a multiple of 16; that must reproduce the reference's per-step stream (distribution.py:67-70) bit for bit."""
    for N, A, T in ((2048, 6, 16), (8, 6, 5), (64, 3, 4)):
        assert (N * A) % 16 == 0
        torch.manual_seed(3)
        per_step = torch.stack([torch.randn(N, A) for _ in range(T)])
        torch.manual_seed(3)
        block = torch.randn(T * N, A).view(T, N, A)
        assert torch.equal(per_step, block), (N, A)


@pytest.mark.parametrize("n", [1 << 16, 128 * 64 * 6 * 4, 16 * 4099])
def test_parallel_reference_noise_equals_one_randn_call(n):
    """torchrl_amd/collector/noise.py: a block of the reference's exploration noise (CPU torch generator,
    torchrl/policies/distribution.py:60-76) drawn by P threads from the engine states trl_mt19937_advance derives is the
    block ONE torch.randn call returns, element for element, and leaves the default generator where that call leaves it --
    from an arbitrary position of the stream, for thread counts that do and do not divide the block."""
    import torch
    from torchrl_amd.collector import noise
    torch.manual_seed(n)
    torch.randn(37)                                                        # somewhere inside a 624-word state block
    start = torch.get_rng_state()
    want = torch.randn(n)
    end, tail = torch.get_rng_state(), torch.randn(5)
    for threads in (2, 3, 8):
        torch.set_rng_state(start)
        got = noise.randn_into(torch.empty(n), threads=threads)
        assert torch.equal(got, want), threads
        assert torch.equal(torch.get_rng_state(), end)
        assert torch.equal(torch.randn(5), tail)
    torch.set_rng_state(start)                                             # 2-D tensors are filled in memory order
    assert torch.equal(noise.randn_into(torch.empty(n // 2, 2), threads=4).view(-1), want)


def test_parallel_reference_noise_falls_back_for_small_or_ragged_blocks():
    import torch
    from torchrl_amd.collector import noise
    for n in (5, 48, noise.MIN_PARALLEL - 16, noise.MIN_PARALLEL + 7):    # below the threshold / not a multiple of 16
        torch.manual_seed(1)
        want = torch.randn(n)
        torch.manual_seed(1)
        assert torch.equal(noise.randn_into(torch.empty(n), threads=4), want)
    with pytest.raises(Exception):
        noise.randn_into(torch.empty(64, dtype=torch.float64))


@pytest.mark.parametrize("world,n_local,a_dim,steps", [(2, 32, 6, 16), (8, 2048, 6, 4), (4, 8, 2, 33)])
def test_sharded_reference_noise_is_the_column_block_of_the_draw_for_all_envs(world, n_local, a_dim, steps):
    """Env shards on several ranks: the reference draws ONE (N_total, A) tensor per vector step
    (torchrl/policies/distribution.py:60-76); rank r's rows [r * N_local, (r + 1) * N_local) of every step come out of
    noise.randn_shard_into without drawing the other ranks' rows, bit for bit, and the default generator ends where the
    draws for ALL envs leave it -- for one and several host threads."""
    import torch
    from torchrl_amd.collector import noise
    n_total = world * n_local
    torch.manual_seed(world)
    torch.randn(21)
    start = torch.get_rng_state()
    whole = torch.stack([torch.randn(n_total, a_dim) for _ in range(steps)])
    end = torch.get_rng_state()
    for threads in (1, 3, 8):
        for r in sorted({0, world // 2, world - 1}):
            torch.set_rng_state(start)
            got = noise.randn_shard_into(torch.empty(steps, n_local, a_dim), steps, n_local, n_total, r * n_local, a_dim,
                                         threads=threads)
            assert torch.equal(got, whole[:, r * n_local:(r + 1) * n_local]), (threads, r)
            assert torch.equal(torch.get_rng_state(), end)
    assert not noise.shard_ok(7, 14, 7, 6)                                 # 42 values per chunk: not a multiple of 16


def test_draw_block_leaves_the_default_generator_alone_and_falls_back_on_an_unknown_torch():
    """draw_block works on private generators from a state handed to it (the prefetch worker's call: the default generator
    belongs to the main thread); when the self-check of the private torch facts fails, randn_into is a plain torch.randn."""
    import torch
    from torchrl_amd.collector import noise
    assert noise._self_check() and noise.fast_path_ok()
    torch.manual_seed(4)
    s0 = torch.get_rng_state()
    n = noise.MIN_PARALLEL * 2
    want = torch.randn(n)
    end = torch.get_rng_state()
    torch.manual_seed(1234)                                                # the default generator is somewhere else entirely
    before = torch.get_rng_state()
    got = torch.empty(n)
    assert torch.equal(noise.draw_block(s0, got, threads=4), end) and torch.equal(got, want)
    assert torch.equal(torch.get_rng_state(), before)
    old, noise._checked = noise._checked, False                            # "another torch build"
    try:
        torch.set_rng_state(s0)
        blocks = noise.STATS["blocks"]
        assert torch.equal(noise.randn_into(torch.empty(n), threads=4), want) and noise.STATS["blocks"] == blocks
        assert not noise.shard_ok(32, 64, 32, 6)
    finally:
        noise._checked = old


def test_native_noise_helper_draws_the_chunks_the_python_threads_draw():
    """include/trl_noise.h (libtrl_noise.so, optional): torch's own normal_() on private generators from plain threads must
    give the chunks the Python-thread path gives -- contiguous segments and a rank's rows of every step -- and exports
    the three symbols its header declares."""
    import ctypes
    import torch
    from torchrl_amd.collector import noise
    lib = noise.native_helper()
    if lib is None:
        pytest.skip("libtrl_noise.so is not built on this machine")
    for name in ("trl_noise_abi_version", "trl_noise_last_error", "trl_noise_draw_chunks"):
        assert hasattr(lib, name)
    assert lib.trl_noise_draw_chunks(None, 5056, 1, None, None, None, 1) != 0 and b"bad arguments" in lib.trl_noise_last_error()
    torch.manual_seed(8)
    s0 = torch.get_rng_state()
    a, b = torch.empty(1 << 17), torch.empty(1 << 17)
    sa, sb = torch.empty(16, 64, 6), torch.empty(16, 64, 6)
    before = noise.STATS["native_blocks"]
    end_a = noise.draw_block(s0, a, threads=4)
    end_sa = noise.draw_block(s0, sa.view(-1), n_chunks=16, stride=4 * 64 * 6, offset=2 * 64 * 6, threads=3)
    assert noise.STATS["native_blocks"] == before + 2
    saved, noise._ext_lib = noise._ext_lib, None                           # the Python-thread path
    try:
        end_b = noise.draw_block(s0, b, threads=4)
        end_sb = noise.draw_block(s0, sb.view(-1), n_chunks=16, stride=4 * 64 * 6, offset=2 * 64 * 6, threads=3)
    finally:
        noise._ext_lib = saved
    assert torch.equal(a, b) and torch.equal(end_a, end_b) and torch.equal(sa, sb) and torch.equal(end_sa, end_sb)
    assert torch.equal(a, torch.randn(1 << 17))                            # (the default generator still stands at s0)


def test_mt19937_advance_matches_the_engine():
    """The state after k engine calls, for k around the 624-word regeneration boundaries, equals the default generator's
    state after a k-element float32 normal_() (one call per element for k >= 16, k % 16 == 0)."""
    import torch
    from torchrl_amd.collector import noise
    torch.manual_seed(99)
    base = torch.get_rng_state()
    for k in (16, 608, 624, 640, 1248, 624 * 5 + 16, 100000 - 100000 % 16):
        torch.set_rng_state(base)
        torch.randn(k)
        want = torch.get_rng_state()
        bounds, states = noise.segment_states(base, k, 1)
        assert bounds == [0, k] and torch.equal(states[1], want), k


def test_product_networks_init_is_the_reference_draw_for_draw(golden):
    """SURVEY 8(a) a15 / VERDICT r04 missing #5: the PRODUCT's networks under the fixture's seed are array_equal to the
    reference's (torchrl/networks/init.py:5-47, base.py:8-107, nets.py:12-49): the 17-64-64-{6,1} pair of net_init.npz
    (basic_init's fan = out_features quirk, 0.1 biases, uniform +-3e-3 heads, logstd = log 0.125), the conv 16/32/64 + fc 512
    net and the orthogonal initialiser of cnn_init.npz; and the conv net computes the reference's forward."""
    import numpy as np
    import torch
    import torchrl.networks as networks
    import torchrl.policies as policies
    assert networks.__name__ == "torchrl_amd.networks"

    def same(mod, g, prefix):
        sd = mod.state_dict()
        want = {k[len(prefix):].replace("__", "."): g[k] for k in g.files if k.startswith(prefix)}
        assert sorted(sd) == sorted(want), (sorted(sd), sorted(want))
        for k, v in sd.items():
            assert v.dtype == torch.float32 and np.array_equal(v.numpy(), want[k]), (prefix, k)

    g = golden("net_init")
    torch.manual_seed(42)                                                  # tests/golden/make_golden.py::case_init
    net = dict(hidden_shapes=[64, 64], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.Tanh)
    pf = policies.GuassianContPolicyBasicBias(input_shape=17, output_shape=6, tanh_action=True, **net)
    vf = networks.Net(input_shape=(17,), output_shape=1, **net)
    same(pf, g, "pf_")
    same(vf, g, "vf_")
    g = golden("cnn_init")
    convs = [[16, [8, 8], [4, 4], [0, 0]], [32, [4, 4], [2, 2], [0, 0]], [64, [3, 3], [1, 1], [0, 0]]]
    torch.manual_seed(43)                                                  # ::case_cnn_init
    qf = networks.Net(output_shape=6, base_type=networks.CNNBase, append_hidden_shapes=[512],
                      activation_func=torch.nn.ReLU, input_shape=(4, 36, 36), hidden_shapes=convs)
    same(qf, g, "cnn_")
    assert qf.base.output_shape == 64 and networks.calc_next_shape((4, 36, 36), convs[0]) == (16, 8, 8)
    with torch.no_grad():
        y = qf(torch.from_numpy(g["input_x"]))
    np.testing.assert_allclose(y.numpy(), g["output_y"], rtol=1e-5, atol=1e-5)
    torch.manual_seed(44)
    mlp = networks.Net(input_shape=(11,), output_shape=3, hidden_shapes=[32, 32], append_hidden_shapes=[],
                       base_type=networks.MLPBase, activation_func=torch.nn.ReLU,
                       init_func=networks.orthogonal_init, net_last_init_func=networks.orthogonal_init)
    same(mlp, g, "ortho_")


def test_last_sample_is_the_last_row_of_the_ring():
    """OnPolicyReplayBufferBase.last_sample (torchrl/replay_buffers/on_policy.py:9-14) reads row `max_size - 1` of every
    requested key -- NOT `_top - 1`: it is the transition written last only when the rollout has just filled the ring
    exactly, which is how on_rl_algo.py:22-33 uses it for the bootstrap value.  Pinned directly."""
    import numpy as np
    from torchrl_amd.replay_buffers.on_policy import OnPolicyReplayBuffer
    N, rows = 3, 4
    buf = OnPolicyReplayBuffer(N * rows, env_nums=N, time_limit_filter=True, device="cpu")
    rs = np.random.RandomState(3)
    written = {}
    for t in range(2 * rows + 2):                                          # wraps twice
        sample = {"next_obs": rs.randn(N, 5).astype(np.float32), "terminals": rs.rand(N, 1) > 0.5,
                  "time_limits": rs.rand(N, 1) > 0.5}
        buf.add_sample(sample)
        written[t % rows] = sample
        got = buf.last_sample(["next_obs", "terminals", "time_limits"])
        assert sorted(got) == ["next_obs", "terminals", "time_limits"]
        for k, v in got.items():
            assert tuple(v.shape)[0] == N
            if rows - 1 in written:
                want = np.asarray(written[rows - 1][k], dtype=np.float32).reshape(N, -1)
                assert np.array_equal(v.cpu().numpy().reshape(N, -1), want), (t, k)
            else:
                assert not v.any()                                         # the ring's last row has not been written yet
        if buf._top == 0:                                                  # ring just filled: the row written last
            assert np.array_equal(got["next_obs"].cpu().numpy(), sample["next_obs"])


def test_mt19937_jump_ahead_gives_the_states_of_the_sequential_walk():
    """trl_mt19937_states_at_mt (csrc/trl_mtjump.cpp): the K engine states of a long stream derived by several host threads,
    all but the first starting from the template JUMPED ahead -- F^J = (x^J mod phi)(F) on MT19937's GF(2)-linear window --
    must be the records of the sequential pass BYTE FOR BYTE (all 5056 bytes of torch's generator state, the 31 dead bits of
    the window included: every jump lands a block early and regenerates), for templates anywhere inside a 624-word block,
    for BASELINE cfg 4's position list (rank 3 of 8: 128 chunks 98 304 calls apart) and for random ones; and the values drawn
    from those states are the reference's: this rank's rows of 128 successive torch.randn(16384, 6) draws."""
    import numpy as np
    import torch
    from torchrl_amd import _C
    from torchrl_amd.collector import noise
    assert _C.lib().trl_mt19937_jump_ready() == 1
    g = torch.Generator()
    rs = np.random.RandomState(5)
    before = noise.STATS["jump_passes"]
    for trial in range(5):
        g.manual_seed(100 + trial)
        if trial:                                                          # trial 0: the freshly seeded state (left = 1, next = 0)
            torch.randn(16 * int(rs.randint(1, 3000)), generator=g)
        s0 = g.get_state()
        if trial < 2:
            S, off, T = 16384 * 6, 3 * 2048 * 6, 128
            pos = [t * S + off for t in range(T)] + [T * S]
        else:
            pos = np.sort(rs.randint(0, 9_000_000, size=50)).tolist()
            pos[0] = 0 if trial == 2 else pos[0]
        seq = noise.states_at(s0, pos)
        for threads in (2, 5, 8):
            assert torch.equal(noise.states_at(s0, pos, threads=threads), seq), (trial, threads)
    assert noise.STATS["jump_passes"] - before == 15
    # end to end at cfg 4's layout (fewer steps): rows [6144, 8192) of T successive (16384, 6) draws
    T, n, n_total, e0, A = 24, 2048, 16384, 3 * 2048, 6
    torch.manual_seed(77)
    want = torch.stack([torch.randn(n_total, A)[e0:e0 + n] for _ in range(T)])
    tail = torch.randn(4)
    torch.manual_seed(77)
    got = noise.randn_shard_into(torch.empty(T, n, A), T, n, n_total, e0, A, threads=4)
    assert torch.equal(got, want) and torch.equal(torch.randn(4), tail)   # values and the generator's end state
    assert noise.STATS["jump_passes"] - before >= 15                       # (16 when the native helper is loaded)
