"""Time the on-policy collection at the benchmark shape with and without the running observation normaliser
(per-step launch sequence vs the persistent rollout kernel).  Development aid."""
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    from torchrl_amd.collector.on_policy import VecOnPolicyCollector
    from torchrl_amd.env.base_wrapper import NormObs
    from torchrl_amd.env.synth import SynthVecEnv
    from torchrl_amd.replay_buffers.on_policy import OnPolicyReplayBuffer
    dev = torch.device("cuda:0")
    agent, col0 = bench.build_agent(dev, 1, 0)
    N, T = bench.N_PER_GPU, bench.T
    out = {}
    for name, wrap in (("fused", False), ("obs_norm", True)):
        env = SynthVecEnv(N, device=dev)
        ev = SynthVecEnv(N, device=dev)
        if wrap:
            env, ev = NormObs(env), NormObs(ev)
        buf = OnPolicyReplayBuffer(N * T, env_nums=N, time_limit_filter=True)
        col = VecOnPolicyCollector(col0.vf, env=env, eval_env=ev, pf=col0.pf, replay_buffer=buf, device=dev,
                                   train_render=False, epoch_frames=N * T, max_episode_frames=1000, eval_episodes=1,
                                   noise_mode="device")
        for _ in range(2):
            col.rollout(T)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            col.rollout(T)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 5
        out[name] = {"ms_per_rollout": ms, "us_per_vector_step": 1e3 * ms / T, "env_steps_per_s": N * T / ms * 1e3}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
