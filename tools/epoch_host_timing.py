"""Host-side breakdown of an off-policy update epoch: time to launch (draw + upload + graph launch / per-update launches),
to wait for the device, and to resolve the info dicts.  argv: dqn | sac  (run from the repo root on a GPU box)."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
which = sys.argv[1]
if which == "dqn":
    import bench_dqn as b
else:
    import bench_sac as b
import torchrl_amd.algo.off_policy.off_rl_algo as m
orig = m.OffRLAlgo.update_per_epoch
T = {"launch": [], "wait": [], "resolve": []}
def patched(self):
    deferred = self.update_deferred
    t0 = time.perf_counter()
    whole = getattr(self, "update_epoch_deferred", None)
    pending = whole(self.opt_times) if whole is not None else None
    if pending is None:
        pending = [deferred(self._sample()) for _ in range(self.opt_times)]
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    infos = self.resolve_updates(pending)
    for info in infos:
        self.logger.add_update_info(info)
    t3 = time.perf_counter()
    T["launch"].append(t1 - t0); T["wait"].append(t2 - t1); T["resolve"].append(t3 - t2)
m.OffRLAlgo.update_per_epoch = patched
sys.argv = [sys.argv[0], "--epochs", "20"]
if which == "dqn":
    b.run(1, 20, False)
else:
    b.main()
for k, v in T.items():
    print(k, "us median %.1f" % (1e6 * float(np.median(v[-15:]))))
