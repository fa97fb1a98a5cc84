"""Full-size (2048 x 128) check of the two update chains against the joint sequence: buffers after every rollout,
parameters at the end.  Development aid."""
import os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import test_fullsize_gpu as T

res = []
for chains in ("joint", "two"):
    os.environ["TRL_PPO_CHAINS"] = chains
    pf, vf, env, buf, col, agent = T.make(T.N, seed=3)
    agent.opt_epochs = 10
    snaps = []
    for e in range(4):
        torch.manual_seed(e)
        col.train_one_epoch()
        agent.current_epoch = e
        np.random.seed(e)
        agent.update_per_epoch()
        snaps.append({k: getattr(buf, "_" + k).clone() for k in ("obs", "values", "rewards", "advs")})
    torch.cuda.synchronize()
    res.append((snaps, agent.engine().flat.clone()))
for e, (a, b) in enumerate(zip(res[0][0], res[1][0])):
    print("epoch", e, {k: float((a[k] - b[k]).abs().max()) for k in a})
print("params", float((res[0][1] - res[1][1]).abs().max()))
