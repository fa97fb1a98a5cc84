"""Within-run A/B of the dominant kernel (ppo_grad_wave_kernel) across builds of the library: for every .so given
on the command line (a child process each, alternating order, `--rounds` times) time the kernel at the benchmark
shape with HIP events and print a checksum of the parameters after the same sequence of updates -- variants that
only re-schedule instructions must reproduce it bit for bit.  Development aid, not part of the product.
    python tools/ab_grad.py torchrl_amd/lib/libtrl_hip_old.so torchrl_amd/lib/libtrl_hip.so"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    import numpy as np
    import torch
    sys.path.insert(0, REPO)
    from torchrl_amd import _C
    _C.LIB_PATH = os.environ["TRL_LIB"]
    import bench
    dev = torch.device("cuda:0")
    agent, col = bench.build_agent(dev, 1, 0)
    col.env.reset()
    col.rollout(col.sample_epoch_frames)
    agent.current_epoch = 0
    np.random.seed(0)
    agent.update_per_epoch()
    eng = agent.engine()
    buf = agent.replay_buffer
    t = {"obs": buf._obs, "acts": buf._acts, "advs": buf._advs, "rets": buf._estimate_returns,
         "old_values": buf._values, "old_logp": buf._old_logp}
    idx = np.random.RandomState(1).permutation(128).reshape(4, 32).astype(np.int64)
    probes = []
    eng.probe = probes
    for _ in range(int(os.environ.get("AB_REPS", "8"))):
        eng.run(t, idx, buf.env_nums)
    torch.cuda.synchronize()
    us = np.array([s.elapsed_time(e) for s, e in probes][4:]) * 1e3
    chk = float(eng.flat.double().abs().sum().item())
    g = float(eng.grads.double().abs().sum().item())
    print("RESULT %s mean %.2f us min %.2f us n=%d  params %.17g grads %.17g"
          % (os.path.basename(os.environ["TRL_LIB"]), us.mean(), us.min(), len(us), chk, g), flush=True)


def main():
    libs = [a for a in sys.argv[1:] if not a.startswith("--")]
    rounds = 2
    for a in sys.argv[1:]:
        if a.startswith("--rounds="):
            rounds = int(a.split("=")[1])
    for r in range(rounds):
        for lib in (libs if r % 2 == 0 else libs[::-1]):
            env = dict(os.environ, TRL_LIB=os.path.abspath(lib), AB_CHILD="1")
            out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
            lines = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
            print(lines[0] if lines else "FAILED %s: %s" % (lib, out.stderr[-800:]), flush=True)


if __name__ == "__main__":
    child() if os.environ.get("AB_CHILD") == "1" else main()
