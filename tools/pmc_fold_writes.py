"""Whose bytes are the WRITE_SIZE that rocprofv3 attributes to the fold kernel?  Per minibatch: the gradient kernel, then
the plain fold (trl_ppo_reduce_f32: reads the 256 partial rows, writes 44 KB of gradients) TWICE in a row.  Under
`rocprofv3 --pmc WRITE_SIZE` the first fold of each pair carries the write-back of the partial rows its predecessor left
dirty in the L2s, the second one only its own stores (profiles/r03_fold_write_attribution.txt).  Development aid."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import ctypes as C
    import bench
    from torchrl_amd import _C
    dev = torch.device("cuda:0")
    agent, col = bench.build_agent(dev, 1, 0)
    col.env.reset()
    col.rollout(col.sample_epoch_frames)
    agent.current_epoch = 0
    np.random.seed(0)
    agent.update_per_epoch()
    agent.logger.drain()
    eng, buf = agent.engine(), agent.replay_buffer
    lib, stream = _C.lib(), _C.stream_ptr(dev)
    K, rows_mb, N = 4, 32, buf.env_nums
    idx = torch.from_numpy(np.random.RandomState(1).permutation(128).reshape(K, rows_mb).astype(np.int64)).to(dev)
    raw = torch.zeros(K, 4, dtype=torch.float64, device=dev)
    _C.adv_stats(buf._advs.reshape(128, N), idx, raw)
    info = torch.zeros(24, dtype=torch.float64, device=dev)
    n_wg, n_pf = eng._n_wg(rows_mb * N)
    g = _C.PpoBatchArgs()
    for k, t in (("obs", buf._obs), ("acts", buf._acts), ("advs", buf._advs), ("rets", buf._estimate_returns),
                 ("old_values", buf._values), ("old_logp", buf._old_logp)):
        setattr(g, k, t.data_ptr())
    g.loss_mode, g.rows_mb, g.N, g.n_global = _C.LOSS_PPO_CLIP, rows_mb, N, float(rows_mb * N)
    g.pf_params, g.vf_params = eng.flat.data_ptr(), eng.flat.data_ptr() + 4 * eng.P_pf
    g.D, g.H, g.A, g.act = eng.D, eng.H, eng.A, eng.act
    g.clip_para, g.entropy_coeff, g.clipped_value_loss, g.tanh_action = 0.2, 0.005, 0, 1
    g.partial, g.scal_partial, g.n_wg, g.n_wg_pf = eng.partial.data_ptr(), eng.scal.data_ptr(), n_wg, n_pf
    fused = "--fused" in sys.argv                        # the fold / clip / Adam launch of the single-process path instead
    a = _C.AdamArgs()
    a.params, a.grads, a.exp_avg, a.exp_avg_sq = eng.flat.data_ptr(), eng.grads.data_ptr(), eng.m.data_ptr(), eng.v.data_ptr()
    a.n_groups = 2
    a.group_sizes[0], a.group_sizes[1] = eng.P_pf, eng.P_vf
    a.group_lr[0], a.group_lr[1] = 0.0, 0.0              # (the measurement must not move the parameters)
    a.max_norm, a.beta1, a.beta2, a.eps, a.grad_scale = 0.5, 0.9, 0.999, 1e-5, 1.0
    a.device_state, a.step_count = 1, 1
    eng._set_device_hyper(0.0, 0.0)
    for rep in range(3):
        for k in range(K):
            g.row_idx, g.adv_raw = idx.data_ptr() + 8 * rows_mb * k, raw.data_ptr() + 32 * k
            _C.check(lib.trl_ppo_minibatch_grad_f32(C.byref(g), stream), "grad")
            for _ in range(2):
                if fused:
                    _C.check(lib.trl_ppo_reduce_adam_f32(eng.partial.data_ptr(), eng.scal.data_ptr(), n_wg, n_pf, eng.D, eng.H,
                                                         eng.A, eng.grads.data_ptr(), info.data_ptr(), C.byref(a),
                                                         eng.red_ws.data_ptr(), stream), "fold+adam")
                else:
                    _C.check(lib.trl_ppo_reduce_f32(eng.partial.data_ptr(), eng.scal.data_ptr(), n_wg, n_pf, eng.D, eng.H, eng.A,
                                                    eng.flat.data_ptr(), eng.grads.data_ptr(), info.data_ptr(), stream), "fold")
    torch.cuda.synchronize()
    print("launched %d x {grad, fold, fold}" % (3 * K))


if __name__ == "__main__":
    main()
