"""Worker of tests/test_dist_gpu.py: one rank of an env-sharded PPO run (or the single-process reference run).
argv: rank world port out_path.  Both ranks use cuda:0; the collectives go through gloo (RCCL refuses two ranks on one
device), which exercises exactly the code the nccl backend runs: sharded envs, C1 / C2 / C3 reductions, the unfused
reduce -> all-reduce -> clip + Adam sequence."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

N_TOTAL, T, ROWS_MB, EPOCHS, HORIZON, MAX_FRAMES = 64, 16, 4, 3, 12, 9
if os.environ.get("TRL_TEST_SIZES"):            # "n_total,T,rows_per_minibatch,epochs": BASELINE cfg 4's real sizes in the 8-rank test
    N_TOTAL, T, ROWS_MB, EPOCHS = (int(v) for v in os.environ["TRL_TEST_SIZES"].split(","))


class Log:
    """Takes launched-but-not-awaited updates like torchrl_amd.utils.Logger: the multi-rank runs go through the deferred
    read of the statistics (the comm check then happens where they are read)."""
    def __init__(self): self.infos, self.later = [], []
    def add_update_info(self, d): self.drain(); self.infos.append(dict(d))
    def add_update_infos_later(self, resolve): self.later.append(resolve)
    def drain(self):
        later, self.later = self.later, []
        for resolve in later:
            self.infos.extend(dict(d) for d in resolve())
    def add_epoch_info(self, *a, **k): pass
    def log(self, *a): pass
    def finish(self): pass


def _lib():
    from torchrl_amd import _C
    return _C.lib()


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    mode = sys.argv[5] if len(sys.argv) > 5 else ""
    noise_opt = sys.argv[6] if len(sys.argv) > 6 else "device"        # device | host | host_prefetch | host_perstep
    obs_norm = mode == "obs_norm"
    if (world > 1 or mode == "nccl_graph") and mode != "nccl_peer":
        import torch.distributed as td
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        if mode == "nccl_graph":           # one rank, RCCL backend, collectives forced on (env set by the test)
            torch.cuda.set_device(0)
            td.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        else:
            td.init_process_group("gloo", rank=rank, world_size=world)
    peer = False
    if mode in ("peer", "nccl_peer", "nccl_graph", "peer_timeout"):
        # the library's own communicator (include/trl_hip.h: trl_comm_*): peer-mapped buffers between the two processes
        # sharing cuda:0 (no RCCL communicator: it refuses two ranks per device), or RCCL + peers on a one-rank group
        if mode == "nccl_peer":
            torch.cuda.set_device(0)
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
            import torch.distributed as td
            td.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        from torchrl_amd import dist as trl_dist
        if mode != "nccl_graph":
            peer = trl_dist.init_comm(torch.device("cuda:0"), use_rccl=(mode == "nccl_peer"))
            assert peer, "peer transport did not come up"
            if os.environ.get("TRL_TEST_WAIT_BLOCKS"):               # the fold launch's resident footprint, overridden
                assert _lib().trl_comm_set_wait_footprint(trl_dist.comm_handle(), int(os.environ["TRL_TEST_WAIT_BLOCKS"])) == 0
            assert bool(_lib().trl_comm_has_rccl(trl_dist.comm_handle())) == (mode == "nccl_peer")
    if mode == "peer_timeout":
        # rank 0 starts an exchange that rank 1 never joins (TRL_COMM_TIMEOUT_S = 1 from the test): the wait must give up, the
        # flag must be readable WITHOUT idling the device (check_comm(peek=True), what the training loop calls once per
        # iteration) and name the missing rank; the flag is cleared by the read
        from torchrl_amd import _C as trl_C
        import torch.distributed as td
        raised, text, again = 0, "", 0
        if rank == 0:
            x = torch.ones(8, device="cuda:0")
            trl_dist.all_reduce_sum_(x)
            torch.cuda.synchronize()
            try:
                trl_dist.check_comm(peek=True)
            except trl_C.TrlError as exc:
                raised, text = 1, str(exc)
            try:
                trl_dist.check_comm()
            except trl_C.TrlError:
                again = 1
        np.savez(out, raised=np.array(raised), text=np.array(text), again=np.array(again))
        td.barrier()
        trl_dist.destroy_comm()
        td.destroy_process_group()
        return
    import torchrl.networks as networks
    import torchrl.policies as policies
    from torchrl.algo import A2C, PPO, TRPO, VMPO
    from torchrl.collector.on_policy import VecOnPolicyCollector
    from torchrl.env.synth import SynthVecEnv
    from torchrl.replay_buffers.on_policy import OnPolicyReplayBuffer
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    np.random.seed(0)                                                  # identical index streams on every rank
    net = dict(hidden_shapes=[64, 64], append_hidden_shapes=[], base_type=networks.MLPBase, activation_func=torch.nn.Tanh)
    pf = policies.GuassianContPolicyBasicBias(input_shape=17, output_shape=6, tanh_action=True, **net)
    vf = networks.Net(input_shape=(17,), output_shape=1, **net)
    n = N_TOTAL // world
    kw = dict(horizon=HORIZON, device=dev, index_offset=rank * n, total_env_nums=N_TOTAL)
    env, eval_env = SynthVecEnv(n, **kw), SynthVecEnv(n, **kw)
    if obs_norm:                                                       # running normaliser shared by all ranks
        from torchrl.env import NormObs
        env, eval_env = NormObs(env), NormObs(eval_env)
    env.seed(3)
    buf = OnPolicyReplayBuffer(n * T, env_nums=n, time_limit_filter=True, device=dev)
    col = VecOnPolicyCollector(vf, env=env, eval_env=eval_env, pf=pf, replay_buffer=buf, device=dev, epoch_frames=n * T,
                               max_episode_frames=MAX_FRAMES, noise_mode="device" if noise_opt == "device" else "host",
                               prefetch_noise=(noise_opt == "host_prefetch"))
    if noise_opt == "host_perstep":        # the reference's literal op sequence: one torch.randn(N_total, A) per vector step
        from torchrl_amd.collector import noise as _noise
        _noise._checked = False
    if obs_norm:
        col.force_per_step = True          # single process: the per-step sequence too (its Philox layout differs from the
    logger = Log()                         # cooperative kernel's), so that the two runs draw the same exploration noise
    common = dict(pf=pf, vf=vf, plr=3e-4, vlr=3e-4, tau=0.95, shuffle=True, entropy_coeff=0.005, discount=0.99, num_epochs=10,
                  batch_size=ROWS_MB * n, gae=True, env=env, replay_buffer=buf, collector=col, logger=logger, device=dev,
                  save_dir=None)
    if mode == "a2c":
        agent = A2C(**common)
    elif mode == "vmpo":                   # replicated update on the gathered minibatch (global top half by advantage)
        agent = VMPO(opt_epochs=2, eta_eps=0.02, alpha_eps=0.1, **common)
    elif mode == "trpo":                   # replicated natural-gradient step on the gathered epoch
        agent = TRPO(max_kl=0.01, cg_damping=1e-2, v_opt_times=2, cg_iters=10, residual_tol=1e-10, **common)
    else:
        agent = PPO(clip_para=0.2, opt_epochs=2, **common)
    if os.environ.get("TRL_TEST_VALUE_CHAIN_DELAY"):                   # hold every value chain back (device spin): the next
        agent.engine()._test_value_chain_delay = int(os.environ["TRL_TEST_VALUE_CHAIN_DELAY"])   # rollout runs beside / ahead of it
    torch.manual_seed(21)                  # the exploration-noise stream of the host modes (the CPU generator)
    acts = []
    for epoch in range(EPOCHS):
        col.rollout(col.sample_epoch_frames)
        acts.append(buf._acts.cpu().numpy().copy())
        agent.current_epoch = epoch
        agent.update_per_epoch()
    logger.drain()
    col.stop_noise_prefetch()
    tail = torch.randn(5).numpy()          # where the CPU stream stands afterwards
    pre = col._prefetcher
    keys = sorted(logger.infos[0])
    np.savez(out, pf=pf.flat_params().cpu().numpy(), vf=vf.flat_params().cpu().numpy(), keys=np.array(keys),
             infos=np.array([[i[k] for k in keys] for i in logger.infos if sorted(i) == keys]),
             obs=buf._obs.cpu().numpy(), rewards=buf._rewards.cpu().numpy(), acts=np.stack(acts), tail=tail,
             prefetched=np.array(0 if pre is None else sum(pre.transport_counts.values()) - pre.dropped_blocks - 1),
             norm_state=env._obs_normalizer.state.cpu().numpy() if obs_norm else np.zeros(1),
             peer=np.array(int(peer)), graph=np.array(int(getattr(agent.engine(), "_graph", None) is not None or bool(getattr(agent.engine(), "_chain_graphs", None)))),
             chains=np.array(int(bool(getattr(agent.engine(), "_chain_graphs", None)))))
    if world > 1 or mode in ("nccl_graph", "nccl_peer"):
        import torch.distributed as td
        from torchrl_amd import dist as trl_dist
        trl_dist.destroy_comm()
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
