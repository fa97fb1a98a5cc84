"""Frame-deduplicating replay.

Parity: the device buffer is driven through the calls the collector makes (begin_episodes / mark_obs_row /
append_step) with the frame events of tests/golden/frame_dedup.npz and must hand back, for every replay row of every
env, the stacks the REFERENCE's FrameStack + LazyFrames + MemoryEfficientReplayBuffer re-encoded
(env/atari_wrapper.py:142-227, replay_buffers/memory_efficient_replay_buffer.py:5-33) -- ring wrap-around, done
resets and over-length resets included; at one env also the reference's own random_batch stream.  The remaining
tests check the collector integration (same batches as the plain ring on the on-GPU frame env), overrun
detection and the footprint."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


FRAME_TAGS = ["done", "surpass", "mixed", "single", "single84"]


def _drive_from_golden(g, tag, **buf_kw):
    """Replays the golden's frame events into the device buffer exactly as VecCollector._step_frames drives it; the
    post-step / post-reset stacks come from oracle.frames.FrameStackOracle (pinned to the reference on the CPU)."""
    from oracle.frames import FrameSourceCPU, FrameStackOracle
    from torchrl.replay_buffers import MemoryEfficientReplayBuffer
    N, rows, steps, H, W, horizon, max_frames, seed = (int(v) for v in g[f"{tag}_args"])
    srcs = [FrameSourceCPU(g[f"{tag}_frames"][n], horizon) for n in range(N)]
    stacks = [FrameStackOracle(4) for _ in range(N)]
    cur = [np.asarray(stacks[n].reset(srcs[n].reset())) for n in range(N)]
    step_cnt = [0] * N
    buf = MemoryEfficientReplayBuffer(N * rows, env_nums=N, device=DEV, **buf_kw)
    up = lambda arrs: torch.tensor(np.stack(arrs), device=DEV)
    buf.begin_episodes(up(cur))
    acts = g[f"{tag}_acts"]
    for t in range(steps):
        row = buf._top
        buf.mark_obs_row()
        rew, done, mask = np.zeros((N, 1), np.float32), np.zeros((N, 1), np.float32), np.zeros(N, np.uint8)
        for n in range(N):
            f, r, d, _ = srcs[n].step(acts[t, n])
            cur[n] = np.asarray(stacks[n].step(f))
            step_cnt[n] += 1
            rew[n], done[n] = r, float(d)
            mask[n] = d or step_cnt[n] >= max_frames
        buf._ensure_key("acts", (N, 1))[row].copy_(torch.tensor(acts[t].astype(np.float32)).view(N, 1))
        buf._ensure_key("rewards", (N, 1))[row].copy_(torch.tensor(rew))
        buf._ensure_key("terminals", (N, 1))[row].copy_(torch.tensor(done))
        buf.append_step(up(cur))                                         # the one new frame of next_obs
        for n in range(N):
            if mask[n]:
                cur[n] = np.asarray(stacks[n].reset(srcs[n].reset()))
                step_cnt[n] = 0
        if mask.any():
            buf.begin_episodes(up(cur), torch.tensor(mask, device=DEV))
        buf._advance()
    return buf


@pytest.mark.parametrize("tag", FRAME_TAGS)
def test_dedup_rows_equal_reference_lazyframes(golden, tag):
    g = golden("frame_dedup")
    N, rows, steps, H, W = (int(v) for v in g[f"{tag}_args"][:5])
    buf = _drive_from_golden(g, tag)
    assert (buf._top, buf._size) == tuple(int(v) for v in g[f"{tag}_top_size"])
    idx = torch.arange(rows, device=DEV)
    for key in ("obs", "next_obs"):
        got = buf._gather(key, idx).cpu().numpy().reshape(rows, N, 4, H, W)
        assert np.array_equal(got, g[f"{tag}_ref_{key}"]), key          # bit-exact bytes
    for key in ("acts", "rewards", "terminals"):
        got = buf._gather(key, idx).cpu().numpy().reshape(rows, N, 1)
        assert np.array_equal(got.astype(np.float64), g[f"{tag}_ref_{key}"]), key
    buf.check_overrun()
    # the trl_gather_rows_u8 twin of the same data: a plain ring filled with the reference's stacks
    from torchrl_amd import _C
    plain = torch.tensor(g[f"{tag}_ref_obs"], device=DEV)
    pick = torch.tensor([rows - 1, 0, rows // 2], device=DEV)
    assert torch.equal(_C.gather_rows(plain, pick).reshape(-1, 4, H, W), buf._gather("obs", pick))


def test_dedup_random_batch_equals_reference_random_batch(golden):
    """One env: B // N = B, so the index stream and the batches are the reference buffer's own
    (memory_efficient_replay_buffer.py:27-33), draw for draw; `out=` destinations are honoured for frame keys."""
    g = golden("frame_dedup")
    buf = _drive_from_golden(g, "single")
    seed = int(g["single_args"][7])
    keys = ["obs", "next_obs", "acts", "rewards", "terminals"]
    np.random.seed(seed + 50)
    for k in range(3):
        if k == 2:                                                       # fixed-address destinations (graph replay)
            out = {"obs": torch.zeros(7, 4, 12, 12, dtype=torch.uint8, device=DEV),
                   "next_obs": torch.zeros(7, 4, 12, 12, dtype=torch.uint8, device=DEV)}
            batch = buf.random_batch(7, keys, out=out)
            assert batch["obs"].data_ptr() == out["obs"].data_ptr() and batch["next_obs"].data_ptr() == out["next_obs"].data_ptr()
        else:
            batch = buf.random_batch(7, keys)
        for key in keys:
            want = g[f"single_batch{k}_{key}"]
            assert np.array_equal(batch[key].cpu().numpy().reshape(want.shape).astype(np.float64), want.astype(np.float64)), (k, key)


def test_dedup_vector_batches_equal_oracle(golden):
    """Several envs: B // N sampled rows x all envs (replay_buffers/base.py:39-51) from the oracle ring."""
    from tests.test_oracle_golden import frame_oracle_from_golden
    g = golden("frame_dedup")
    for tag in ("done", "mixed"):
        buf, ring = _drive_from_golden(g, tag), frame_oracle_from_golden(g, tag)
        N = int(g[f"{tag}_args"][0])
        for k in range(3):
            np.random.seed(300 + k)
            idx, want = ring.random_batch(2 * N, ["obs", "next_obs", "rewards"])
            np.random.seed(300 + k)
            got = buf.random_batch(2 * N, ["obs", "next_obs", "rewards"])
            for key in want:
                assert np.array_equal(got[key].cpu().numpy().astype(np.float64), want[key]), (tag, k, key)


def _run(buf_cls, N, rows, steps, horizon, max_frames, seed, **buf_kw):
    from torchrl.collector import VecCollector
    from torchrl.env import get_vec_env
    from torchrl.policies import EpsilonGreedyDQNDiscretePolicy
    from tests.test_dqn_gpu import small_qnet
    A = 6
    env = get_vec_env("SynthAtari-v0", {"reward_scale": 1}, N)
    eval_env = get_vec_env("SynthAtari-v0", {"reward_scale": 1}, N)
    env.horizon = eval_env.horizon = horizon
    env.seed(seed)
    torch.manual_seed(3)
    qf = small_qnet(A).to(DEV)
    pf = EpsilonGreedyDQNDiscretePolicy(qf=qf, start_epsilon=0.9, end_epsilon=0.5, decay_frames=1000, action_shape=A)
    buf = buf_cls(N * rows, env_nums=N, **buf_kw)
    col = VecCollector(env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=DEV, epoch_frames=N * steps,
                       max_episode_frames=max_frames, eval_episodes=1)
    np.random.seed(seed + 10)
    res = col.train_one_epoch()
    return buf, res


@pytest.mark.parametrize("horizon,max_frames,min_ep", [(5, 1000, 5), (1000, 3, 3), (7, 4, None)])
def test_dedup_batches_equal_plain_ring(horizon, max_frames, min_ep):
    from torchrl.replay_buffers import BaseReplayBuffer, MemoryEfficientReplayBuffer
    N, rows, steps, seed = 8, 6, 17, 4                                       # 17 steps through a 6-row ring
    plain, r1 = _run(BaseReplayBuffer, N, rows, steps, horizon, max_frames, seed)
    dedup, r2 = _run(MemoryEfficientReplayBuffer, N, rows, steps, horizon, max_frames, seed, min_episode_frames=min_ep)
    assert r1["train_epoch_reward"] == r2["train_epoch_reward"]
    assert not hasattr(dedup, "_obs") and not hasattr(dedup, "_next_obs")    # no stacks are stored
    keys = ["obs", "next_obs", "acts", "rewards", "terminals"]
    for k in range(5):
        np.random.seed(100 + k)
        a = plain.random_batch(N * 3, keys)
        np.random.seed(100 + k)
        b = dedup.random_batch(N * 3, keys)
        for key in keys:
            assert a[key].dtype == b[key].dtype and torch.equal(a[key], b[key]), (k, key)
    # every row, both frame keys, byte for byte
    idx = torch.arange(rows, device=DEV)
    assert torch.equal(dedup._gather("obs", idx), plain._obs.reshape(rows * N, 4, 84, 84))
    assert torch.equal(dedup._gather("next_obs", idx), plain._next_obs.reshape(rows * N, 4, 84, 84))
    dedup.check_overrun()


def test_overrun_is_detected_not_silent():
    from torchrl.replay_buffers import MemoryEfficientReplayBuffer
    from torchrl_amd import _C
    # promise 1000-frame episodes, deliver resets every 2 steps: the stream wraps over live frames
    buf, _ = _run(MemoryEfficientReplayBuffer, 4, 12, 30, 1000, 2, 1, min_episode_frames=1000)
    np.random.seed(0)
    buf.random_batch(4 * 12, ["obs"])
    buf._gather("obs", torch.arange(12, device=DEV))
    with pytest.raises(_C.TrlError, match="overrun"):
        buf.check_overrun()


def test_dqn_trains_from_dedup_buffer_and_footprint():
    """cfg 5 shape at reduced size: DQN updates from the frame-deduplicating buffer; HBM footprint ratio."""
    from torchrl.algo import DQN
    from torchrl.replay_buffers import BaseReplayBuffer, MemoryEfficientReplayBuffer
    from tests.test_dqn_gpu import small_qnet
    N, rows = 16, 24
    buf, _ = _run(MemoryEfficientReplayBuffer, N, rows, rows + 5, 1000, 999, 2, min_episode_frames=1000)
    plain, _ = _run(BaseReplayBuffer, N, rows, rows + 5, 1000, 999, 2)
    ratio = (plain._obs.numel() + plain._next_obs.numel()) / buf._stream.numel()
    assert ratio > 5.0, ratio                # 24-row ring: 36 slots vs 192 frames; cfg 5 (195 rows): 207 vs 1560 = 7.5x
    assert buf.footprint_bytes() < 0.2 * sum(getattr(plain, "_" + k).numel() * getattr(plain, "_" + k).element_size()
                                               for k in plain._keys)

    class Stub:
        epoch_frames = 0

    class Log:
        def add_update_info(self, d): pass
        def add_epoch_info(self, *a, **k): pass
        def log(self, *a): pass
        def finish(self): pass
    from torchrl.env import get_vec_env
    from torchrl.policies import EpsilonGreedyDQNDiscretePolicy
    torch.manual_seed(0)
    qf = small_qnet(6).to(DEV)
    pf = EpsilonGreedyDQNDiscretePolicy(qf=qf, start_epsilon=0.9, end_epsilon=0.5, decay_frames=1000, action_shape=6)
    agent = DQN(pf=pf, qf=qf, qlr=1e-3, env=get_vec_env("SynthAtari-v0", {"reward_scale": 1}, N), replay_buffer=buf,
                collector=Stub(), logger=Log(), discount=0.99, num_epochs=1, batch_size=N * 4, device=torch.device(DEV),
                save_dir=None, tau=0.005, use_soft_update=True, opt_times=1)
    before = [p.detach().clone() for p in qf.parameters()]
    for k in range(3):
        np.random.seed(k)
        info = agent.update(buf.random_batch(N * 4, agent.sample_key))
    assert np.isfinite(list(info.values())).all()
    assert any((a - b.detach()).abs().max().item() > 0 for a, b in zip(before, qf.parameters()))
    buf.check_overrun()
