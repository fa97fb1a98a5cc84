"""Within-run A/B of the one-launch minibatch step (trl_ppo_minibatch_step_f32) across builds of the library (tools/mkvariant.sh):
per .so a child process times the step launch at the benchmark shape with HIP events, prints a checksum of the parameters after
the same sequence of updates, and -- for builds with -DSTEP_CLK -- the phase stamps of the last launch (100 MHz wall clock,
us after the earliest workgroup's start; mean / max over the 256 workgroups).  Development aid, not part of the product.
    python tools/ab_step.py torchrl_amd/lib/libtrl_hip_a.so torchrl_amd/lib/libtrl_hip_b.so [--rounds=2] [--split]"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PHASES = ["start", "pass done", "flag up", "all flags", "acquired", "folded", "norms", "end"]


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
    print("RESULT %s one_launch=%s mean %.2f us min %.2f us n=%d  params %.17g err %d"
          % (os.path.basename(os.environ["TRL_LIB"]), eng.one_launch, us.mean(), us.min(), len(us), chk,
             int(eng.red_ws[:1].view(torch.int32).item())), flush=True)
    base = _C.lib().trl_ppo_reduce_adam_workspace(eng.D, eng.H, eng.A)
    nb = (base - 16) // 4
    words = base + 4 + 256 + 4 * nb
    if eng.one_launch and eng.red_ws.numel() >= words + 2 * 8 * 256:
        clk = eng.red_ws[words:words + 2 * 8 * 256].view(torch.int64).view(256, 8).cpu().numpy().astype(np.float64)
        clk = (clk - clk[:, 0].min()) / 100.0
        for k, name in enumerate(PHASES):
            print("   %-10s mean %7.2f  max %7.2f  min %7.2f us" % (name, clk[:, k].mean(), clk[:, k].max(), clk[:, k].min()), flush=True)
        n_wg, n_pf = eng._n_wg(idx.shape[1] * buf.env_nums)
        tiles = idx.shape[1] * buf.env_nums // 16
        n_vf_waves = 4 * (n_wg - n_pf)
        long_vf = (tiles - (tiles // n_vf_waves) * n_vf_waves + 3) // 4       # value workgroups whose waves run one more tile
        d = clk[:, 1] - clk[:, 0]                                            # the pass of each workgroup
        for name, sl in (("policy (%d wgs, %.2f tiles/wave)" % (n_pf, tiles / (4.0 * n_pf)), slice(0, n_pf)),
                         ("value, long (%d wgs, %d tiles)" % (long_vf, tiles // n_vf_waves + 1), slice(n_pf, n_pf + long_vf)),
                         ("value, short (%d wgs, %d tiles)" % (n_wg - n_pf - long_vf, tiles // n_vf_waves), slice(n_pf + long_vf, n_wg))):
            print("   pass of %-36s mean %6.2f  max %6.2f  min %6.2f us" % (name, d[sl].mean(), d[sl].max(), d[sl].min()), flush=True)


def main():
    libs = [a for a in sys.argv[1:] if not a.startswith("--")]
    rounds = 2
    for a in sys.argv[1:]:
        if a.startswith("--rounds="):
            rounds = int(a.split("=")[1])
    for r in range(rounds):
        for lib in (libs if r % 2 == 0 else libs[::-1]):
            env = dict(os.environ, TRL_LIB=os.path.abspath(lib), AB_CHILD="1")
            if "--split" in sys.argv:
                env["TRL_PPO_STEP"] = "split"
            out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
            lines = [l for l in out.stdout.splitlines() if l.startswith("RESULT") or l.startswith("   ")]
            print("\n".join(lines) if lines else "FAILED %s: %s" % (lib, out.stderr[-800:]), flush=True)


if __name__ == "__main__":
    child() if os.environ.get("AB_CHILD") == "1" else main()
