"""Multi-GPU glue: one process per GPU, envs sharded by index.

The reference has no distributed path.  What makes sharding exact here is its
minibatch definition -- `B // N` time rows x ALL N envs
(torchrl/replay_buffers/on_policy.py:75-88): when every rank draws the same row
indices (same numpy seed) and owns envs [rank*N/G, (rank+1)*N/G), the union of
the rank-local minibatches IS the single-process minibatch.  So the data path
needs no collective; only three small reductions exist:
  C1  flat gradient SUM (per-sample grads already carry 1/n_global),
  C2  advantage statistics {sum, sumsq} SUM and {max, -min} MAX, once per epoch,
  C3  logging statistics, once per epoch.

Transport.  `torch.distributed` provides the rendezvous (process group "nccl" ==
RCCL on ROCm, "gloo" in CPU tests).  After `init_comm()` the collectives themselves
go through the library's own C ABI (include/trl_hip.h, csrc/k_comm.hip): a
`trl_comm_t` holding an RCCL communicator for bandwidth-class messages and
peer-mapped (hipIpc / xGMI) buffers for the latency-class ones -- the 44 KB PPO
gradient is exchanged INSIDE the fold/clip/Adam launch, the statistics vectors by
a one-kernel all-reduce; both are ordinary kernel launches, so the whole
multi-rank update sequence replays as one HIP graph.  Without `init_comm()` (CPU
tests, or when the peer set-up fails its self-check) the same calls fall back to
`torch.distributed` collectives.
"""
import ctypes as C
import os

import torch
import torch.distributed as td

# TRL_FORCE_COLLECTIVES=1 issues the collectives even at world size 1 (single-GPU smoke test of the
# call path: dtypes, in-place views, stream ordering)
_FORCE = os.environ.get("TRL_FORCE_COLLECTIVES", "0") == "1"

_comm = None            # ctypes handle of the trl_comm_t
_peer_ok = False        # peers mapped AND the self-check passed on every rank
_has_rccl = False
_peer_report = {}       # what init_comm found: self-check result, ranks per device, why the peers are (not) used
# The peer transport waits INSIDE kernels (a rank's fold launch polls until every rank's granules have arrived), so every
# rank's launches must be able to run at the same time.  One rank per GPU: always.  Ranks SHARING a GPU (the one-GPU
# tests): the waiting fold launches of all of them together must leave CUs for the gradient kernels they are waiting for --
# at the kernel's own grid (180 blocks x 8 waves) two ranks already leave no SIMD with the 416 free registers a gradient
# wave needs on 104 of the 256 CUs, and a third rank starves (round 5).  So the fold launch's resident footprint is a
# property of the communicator (trl_comm_set_wait_footprint): with r > 2 ranks per device every rank's fold launch is
# CUs // (2 r) blocks, all ranks' waiting blocks together cover at most half of the device.
MAX_PEER_RANKS_PER_DEVICE = 16                  # (= trl_comm_max_ranks(): no limit of its own any more)
_SMALL_CAP = 4096       # TRL_XR_CAP_SMALL: 32-bit words per message of the peer transport (statistics region)
_GRAD_CAP = 12288       # TRL_XR_CAP_GRAD: floats per message of its gradient region


def initialized():
    return td.is_available() and td.is_initialized()


def world_size():
    return td.get_world_size() if initialized() else 1


def rank():
    return td.get_rank() if initialized() else 0


def shard(total_envs, world=None, r=None):
    """(offset, count) of the env-index block owned by rank r."""
    world = world_size() if world is None else world
    r = rank() if r is None else r
    if total_envs % world != 0:
        raise ValueError("total env count %d is not divisible by world size %d" % (total_envs, world))
    per = total_envs // world
    return r * per, per


def _active():
    return initialized() and (world_size() > 1 or _FORCE)


def collectives_active():
    """True when gradients / statistics are exchanged between ranks (world size > 1, or forced)."""
    return _active()


# ------------------------------------------------------------------ the library's own communicator
def comm_handle():
    return _comm


def peer_ready():
    """True when the latency transport (peer-mapped buffers) is usable on every rank."""
    return _comm is not None and _peer_ok


def transport():
    """Which route the small exchanges (C1-C3) take: "peer" (hipIpc / xGMI-mapped buffers, plain kernel launches), "rccl"
    (the library's RCCL communicator; torch.distributed's one while a stream captures), "torch.distributed:<backend>", or
    "none"."""
    if not _active():
        return "none"
    if peer_ready():
        return "peer"
    if _comm is not None and _has_rccl:
        return "rccl"
    return "torch.distributed:%s" % td.get_backend()


def _all_agree(flag):
    votes = [None] * world_size()
    td.all_gather_object(votes, bool(flag))
    return all(votes)


def init_comm(device=None, use_rccl=None, peers=True, allow_shared_device=False):
    """Create the trl_comm_t of this process group (idempotent).  `use_rccl` defaults to "the process group is
    nccl" (several ranks per device, as in the single-GPU tests, cannot form an RCCL communicator).  Every step is
    voted on by all ranks, so all of them end up on the same transport.  Returns peer_ready().
    allow_shared_device: keep the peers although more than MAX_PEER_RANKS_PER_DEVICE ranks share a GPU -- only for callers
    whose launches all stay small (the stand-alone all-reduces: tests of the 8-slot buffer layout on one GPU)."""
    global _comm, _peer_ok, _has_rccl
    if _comm is not None or not initialized():
        return peer_ready()
    from . import _C
    lib = _C.lib()
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    torch.cuda.set_device(dev)
    w, r = world_size(), rank()
    if w > lib.trl_comm_max_ranks():
        return False
    if use_rccl is None:
        use_rccl = td.get_backend() == "nccl"
    uid = None
    if use_rccl:
        payload = [None]
        if r == 0:
            buf = C.create_string_buffer(lib.trl_comm_unique_id_bytes())
            payload = [buf.raw if lib.trl_comm_get_unique_id(buf) == 0 else None]
        td.broadcast_object_list(payload, src=0)
        uid = payload[0]
    handle = C.c_void_p()
    rc = lib.trl_comm_init(C.byref(handle), r, w, uid)
    if not _all_agree(rc == 0):
        if rc == 0:
            lib.trl_comm_destroy(handle)
        return False
    _comm, _has_rccl = handle, bool(lib.trl_comm_has_rccl(handle))
    if peers and os.environ.get("TRL_NO_PEER") != "1":
        nb = lib.trl_comm_peer_handle_bytes()
        mine = C.create_string_buffer(nb)
        ok = lib.trl_comm_peer_export(handle, mine) == 0
        gathered = [None] * w
        td.all_gather_object(gathered, (ok, mine.raw))
        if all(g[0] for g in gathered):
            blob = C.create_string_buffer(b"".join(g[1] for g in gathered), nb * w)
            ok = lib.trl_comm_peer_open(handle, blob) == 0
        else:
            ok = False
        if _all_agree(ok):
            _peer_ok = True
            _peer_ok = _all_agree(_self_check(dev))
        _peer_report["self_check"] = "passed on every rank (%d slots per buffer)" % w if _peer_ok else "failed or unavailable"
        # ranks per physical device (host name + PCI bus id): see MAX_PEER_RANKS_PER_DEVICE
        try:
            import socket
            props = torch.cuda.get_device_properties(dev)
            key = (socket.gethostname(), getattr(props, "pci_bus_id", dev.index), getattr(props, "pci_device_id", 0),
                   getattr(props, "pci_domain_id", 0), str(getattr(props, "uuid", "")))
        except Exception:                                                  # noqa: BLE001
            key = ("?", dev.index)
        keys = [None] * w
        td.all_gather_object(keys, key)
        sharing = max(keys.count(k) for k in keys)
        _peer_report["ranks_per_device"] = sharing
        _peer_report["peer_buffer"] = {1: "uncached device memory (hipDeviceMallocUncached)",
                                       0: "plain hipMalloc memory (the uncached allocation failed: see stderr)"}.get(
            int(lib.trl_comm_peer_buffer_kind(handle)), "none")
        if _peer_ok and sharing > 2:                                       # (two ranks at the kernel's own grid leave 76 CUs: round 4)
            cus = torch.cuda.get_device_properties(dev).multi_processor_count
            blocks = max(1, cus // (2 * sharing))
            lib.trl_comm_set_wait_footprint(handle, blocks)
            _peer_report["wait_footprint"] = "%d blocks x 8 waves per rank while a gradient is outstanding (%d ranks on a %d-CU device)" \
                % (blocks, sharing, cus)
        elif _peer_ok:
            _peer_report["wait_footprint"] = "the fold launch's own grid (one block per 64 parameters; %d rank(s) per device)" % sharing
        if _peer_ok and sharing > MAX_PEER_RANKS_PER_DEVICE and not allow_shared_device:
            _peer_ok = False
            _peer_report["not_used"] = ("%d ranks share one device: the in-kernel waits of the peer transport need every "
                                        "rank's launches co-resident (at most %d ranks per device)"
                                        % (sharing, MAX_PEER_RANKS_PER_DEVICE))
        if not _peer_ok and ok:
            lib.trl_comm_peer_enable(handle, 0)
    return peer_ready()


def ranks_per_device():
    """How many ranks share the most crowded physical device (1 on a one-process-per-GPU node; tests put several on one)."""
    return int(_peer_report.get("ranks_per_device", 1))


def peer_report():
    """What init_comm established about the peer transport (for bench.py's `config`)."""
    return dict(_peer_report)


def link_preflight(devices):
    """Pre-flight of a one-node multi-GPU run: for every ordered pair of the given device indices whether peer access is
    possible (hipDeviceCanAccessPeer) and over what (hipExtGetLinkTypeAndHopCount: xGMI or PCIe, hops) -- the peer
    transport pushes granules straight into the other GPUs' memory, RCCL rings run over the same links.  Returns
    {"pairs": n, "peer_access_all": bool, "links": {"xgmi": k, "pcie": m, ...}, "max_hops": h, "no_access": [...]}."""
    from . import _C
    lib = _C.lib()
    names = {0: "hypertransport", 1: "qpi", 2: "pcie", 3: "infiniband", 4: "xgmi"}       # hsa_amd_link_info_type_t
    out = {"pairs": 0, "peer_access_all": True, "links": {}, "max_hops": 0, "no_access": []}
    devs = sorted(set(int(d) for d in devices))
    for a in devs:
        for b in devs:
            if a == b:
                continue
            info = (C.c_int32 * 3)()
            rc = lib.trl_comm_link_info(a, b, info)
            out["pairs"] += 1
            if rc != 0 or not info[0]:
                out["peer_access_all"] = False
                out["no_access"].append([a, b])
                continue
            name = names.get(int(info[1]), "type%d" % info[1]) if info[1] >= 0 else "unknown"
            out["links"][name] = out["links"].get(name, 0) + 1
            out["max_hops"] = max(out["max_hops"], int(info[2]))
    return out


def pin_rank_cpus(local_rank, local_world, reserve=0):
    """Give every local rank its own slice of the host's cores (this process and the threads it starts later: the
    reference-noise draw runs `noise.default_threads()` host threads per rank, one rollout ahead of the device -- eight
    ranks' pools must not migrate over each other).  Slices are contiguous blocks of the CPUs this process may run on,
    so that a rank's threads share a cache complex.  Returns the CPU list, or None when the platform cannot pin."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return None
    per = (len(cpus) - reserve) // max(1, int(local_world))
    if per < 2 or local_world <= 1:
        return None
    mine = cpus[reserve + local_rank * per: reserve + (local_rank + 1) * per]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    return mine


def _self_check(dev):
    """The peer transport against closed-form sums, both buffer halves, every supported element kind."""
    from . import _C
    lib = _C.lib()
    w, r = world_size(), rank()
    stream = _C.stream_ptr(dev)
    try:
        for rep in range(4):
            n = 3000 + rep
            x = (torch.arange(n, device=dev, dtype=torch.float32) % 97 + 1) * (r + 1 + rep)
            if lib.trl_allreduce_sum_f32(x.data_ptr(), n, _comm, stream) != 0:
                return False
            want = (torch.arange(n, device=dev, dtype=torch.float32) % 97 + 1) * sum(q + 1 + rep for q in range(w))
            d = torch.arange(1500, device=dev, dtype=torch.float64).view(-1, 4) * 0 + (r + 1.5 + rep)
            d[:, 3] = -(r + 0.25)
            if lib.trl_allreduce_f64(d.data_ptr(), d.numel(), 4, 0xC, _comm, stream) != 0:
                return False
            g = (torch.arange(11085, device=dev, dtype=torch.float32) % 61 - 30) * (0.5 * (r + 1) + rep)   # gradient region
            if lib.trl_allreduce_sum_f32(g.data_ptr(), g.numel(), _comm, stream) != 0:
                return False
            g_want = (torch.arange(11085, device=dev, dtype=torch.float32) % 61 - 30) * sum(0.5 * (q + 1) + rep for q in range(w))
            torch.cuda.synchronize(dev)
            if lib.trl_comm_error(_comm) != 0 or not torch.equal(x, want) or not torch.equal(g, g_want):
                return False
            if not (bool((d[:, :2] == sum(q + 1.5 + rep for q in range(w))).all()) and bool((d[:, 2] == w - 1 + 1.5 + rep).all())
                    and bool((d[:, 3] == -0.25).all())):
                return False
        return True
    except Exception:
        return False


def destroy_comm():
    global _comm, _peer_ok, _has_rccl
    if _comm is not None:
        from . import _C
        _C.lib().trl_comm_destroy(_comm)
    _comm, _peer_ok, _has_rccl = None, False, False


def check_comm(peek=False):
    """Raise if a peer wait timed out since the last check (a rank went missing); one small synchronous read -- which waits
    for the device to go idle.  `peek=True` reads the flag on a stream of the communicator's own instead, without that wait
    (the per-iteration check of an update whose statistics the caller has already waited for)."""
    if _comm is not None and _peer_ok:
        from . import _C
        if (_C.lib().trl_comm_error_peek(_comm) if peek else _C.lib().trl_comm_error(_comm)) != 0:
            raise _C.TrlError("cross-rank exchange timed out: a rank did not deliver its contribution (%s)"
                              % (comm_error_detail() or "no detail recorded"))


def comm_error_detail():
    """Which contribution the first timed-out peer wait was missing, as text ("" when nothing is recorded)."""
    if _comm is None:
        return ""
    from . import _C
    out = (C.c_int32 * 4)()
    if _C.lib().trl_comm_error_detail(_comm, out) != 0 or out[0] == 0:
        return ""
    return ("rank %d waited for rank %d's %s granules of exchange %d and found tag %d"
            % (rank(), out[1], "gradient" if out[0] == 1 else "statistics", out[2], out[3]))


def _via_abi(t, period, max_mask):
    """Route one in-place reduction through the C ABI when it can carry it; False = use torch.distributed."""
    if _comm is None or not t.is_cuda or not t.is_contiguous():
        return False
    from . import _C
    lib = _C.lib()
    n = t.numel()
    stream = _C.stream_ptr(t.device)
    # the peer transport is plain kernel launches (capturable); the library's RCCL communicator is only used outside
    # stream capture -- captured RCCL calls go through torch.distributed's communicator, the one bench.py probes
    rccl_ok = _has_rccl and not torch.cuda.is_current_stream_capturing()
    if t.dtype == torch.float32 and max_mask == 0 and ((_peer_ok and n <= _GRAD_CAP) or rccl_ok):
        _C.check(lib.trl_allreduce_sum_f32(t.data_ptr(), n, _comm, stream), "trl_allreduce_sum_f32")
        return True
    if t.dtype == torch.float64 and ((_peer_ok and 2 * n <= _SMALL_CAP) or (rccl_ok and max_mask == 0)):
        _C.check(lib.trl_allreduce_f64(t.data_ptr(), n, int(period), int(max_mask), _comm, stream), "trl_allreduce_f64")
        return True
    return False


def all_reduce_sum_(t):
    if _active() and not _via_abi(t, 1, 0):
        td.all_reduce(t, op=td.ReduceOp.SUM)
    return t


def all_reduce_max_(t):
    if _active() and not _via_abi(t, 1, 1):
        td.all_reduce(t, op=td.ReduceOp.MAX)
    return t


def reduce_adv_raw_(raw):
    """raw: (K, 4) float64 {sum, sumsq, max, -min} -> global statistics (C2)."""
    if _active() and not _via_abi(raw, 4, 0xC):
        s = raw[:, :2].contiguous()
        m = raw[:, 2:].contiguous()
        all_reduce_sum_(s)
        all_reduce_max_(m)
        raw[:, :2] = s
        raw[:, 2:] = m
    return raw


# columns of the per-update info rows (include/trl_hip.h, trl_ppo_reduce_f32)
# sums: surrogate, log-prob sum / sum of squares, value loss, value-prediction sum / sum of squares;
# maxima: max / -min of log-prob, ratio and value prediction.  Slots 8..11 and 16..19 derive from logstd (replicated).
INFO_SUM_COLS = [0, 1, 2, 7, 12, 13]
INFO_MAX_COLS = [3, 4, 5, 6, 14, 15]
_INFO_MAX_MASK = sum(1 << c for c in range(24) if c not in INFO_SUM_COLS)     # replicated / unused columns: MAX is idempotent


def reduce_info_(info):
    """info: (K, 24) float64 per-update statistics -> global (C3)."""
    if _active() and not (info.shape[-1] == 24 and _via_abi(info, 24, _INFO_MAX_MASK)):
        s = info[:, INFO_SUM_COLS].contiguous()
        m = info[:, INFO_MAX_COLS].contiguous()
        all_reduce_sum_(s)
        all_reduce_max_(m)
        info[:, INFO_SUM_COLS] = s
        info[:, INFO_MAX_COLS] = m
    return info


def all_gather_cat(t):
    """Rows of every rank's tensor, concatenated in rank order (identity at world size 1)."""
    if not _active():
        return t
    parts = [torch.empty_like(t) for _ in range(world_size())]
    td.all_gather(parts, t.contiguous())
    return torch.cat(parts, dim=0)


def gather_env_shards(t, n_local):
    """(rows * n_local, feat) minibatch shard of every rank -> the (rows * n_total, feat) minibatch a single process
    would hold: row-major over (sampled time row, env), envs in rank order.  Identity at world size 1."""
    if not _active():
        return t
    feat = t.shape[1:]
    rows = t.shape[0] // n_local
    parts = [torch.empty_like(t) for _ in range(world_size())]
    td.all_gather(parts, t.contiguous())
    whole = torch.cat([p.view(rows, n_local, *feat) for p in parts], dim=1)
    return whole.reshape(rows * whole.shape[1], *feat).contiguous()


def shard_rows_of_global(make_global, rows, n_local, feat, device):
    """The (rows * n_local, feat) block of this rank out of a tensor generated for ALL ranks: `make_global(rows * n_total,
    feat)` is laid out (rows, n_total, feat) -- a sampled time row x all envs -- and the rank owns envs
    [rank * n_local, (rank + 1) * n_local).  Every rank generates the same global tensor (same seeds), so the union
    over ranks is exactly what a single process would have drawn."""
    w, r = world_size(), rank()
    if w == 1:
        return make_global(rows * n_local, feat)
    g = make_global(rows * n_local * w, feat).view(rows, n_local * w, feat)
    return g[:, r * n_local:(r + 1) * n_local, :].reshape(rows * n_local, feat).to(device).contiguous()
