"""Development aid: per-launch time of the dense-layer GEMMs (k_gemm.hip) at the SAC / DQN shapes, timed
with HIP events over back-to-back launches.  TRL_LIB=<path> selects an experimental build."""
import sys, os, json
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from torchrl_amd import _C

SHAPES = [(4096, 29, 256), (4096, 256, 256), (4096, 256, 1), (4096, 256, 24), (4096, 64, 256), (4096, 128, 256),
          (4096, 512, 256), (8192, 256, 256), (512, 3136, 512), (512, 512, 6), (512, 512, 1200),
          (204800, 256, 16), (41472, 256, 32), (25088, 288, 64)]


def timed(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    out = []
    for M, K, N in SHAPES:
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
        y = _C.linear_fwd(x, w, b, 1)
        dy = torch.randn(M, N, device=dev)
        ws = torch.empty(_C.lib().trl_linear_bwd_weight_workspace(M, K, N), device=dev)
        dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
        t_f = timed(lambda: _C.linear_fwd(x, w, b, 1))
        t_i = timed(lambda: _C.linear_bwd_input(dy, y, 1, w))
        t_w = timed(lambda: _C.linear_bwd_weight(dy, y, 1, x, dw=dw, db=db, workspace=ws))
        fl = 2.0 * M * K * N
        out.append(dict(M=M, K=K, N=N, fwd_us=round(t_f, 2), bwd_in_us=round(t_i, 2), bwd_w_us=round(t_w, 2),
                        fwd_tf=round(fl / t_f * 1e-6, 1), bwd_in_tf=round(fl / t_i * 1e-6, 1), bwd_w_tf=round(fl / t_w * 1e-6, 1)))
        print(json.dumps(out[-1]), flush=True)


def grouped():
    """Per-launch time and MFMA rate of the grouped dense-layer launches and of the conv kernels at cfg 3 / cfg 5 shapes."""
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for G, M, K, N in ((1, 4096, 256, 256), (2, 4096, 256, 256), (6, 4096, 256, 256), (6, 4096, 23, 256), (6, 4096, 256, 1)):
        xs = [torch.randn(M, K, device=dev) for _ in range(G)]
        ws = [torch.randn(N, K, device=dev) * 0.05 for _ in range(G)]
        bs = [torch.randn(N, device=dev) for _ in range(G)]
        ys = _C.linear_fwd_group(xs, ws, bs, 1)
        dys = [torch.randn(M, N, device=dev) for _ in range(G)]
        dws = [torch.empty(N, K, device=dev) for _ in range(G)]; dbs = [torch.empty(N, device=dev) for _ in range(G)]
        wsp = torch.empty(G * _C.lib().trl_linear_bwd_weight_workspace(M, K, N), device=dev)
        fl = 2.0 * G * M * K * N
        t = [timed(lambda: _C.linear_fwd_group(xs, ws, bs, 1)), timed(lambda: _C.linear_bwd_input_group(dys, ys, 1, ws)),
             timed(lambda: _C.linear_bwd_weight_group(dys, ys, 1, xs, dws, dbs, workspace=wsp))]
        print(json.dumps(dict(G=G, M=M, K=K, N=N, fwd_us=round(t[0], 1), bwd_in_us=round(t[1], 1), bwd_w_us=round(t[2], 1),
                              fwd_tf=round(fl / t[0] * 1e-6, 1), bwd_in_tf=round(fl / t[1] * 1e-6, 1),
                              bwd_w_tf=round(fl / t[2] * 1e-6, 1))), flush=True)
    B = 512
    frames = torch.randint(0, 256, (B, 4, 84, 84), dtype=torch.uint8, device=dev)
    convs = [("conv1 u8 direct", 4, 84, 8, 4, 16), ("conv2 nhwc", 16, 20, 4, 2, 32), ("conv3 nhwc", 32, 9, 3, 1, 64)]
    x = None
    for name, C, H, k, s, Co in convs:
        w = torch.randn(Co, C * k * k, device=dev) * 0.05; b = torch.randn(Co, device=dev)
        Ho = (H - k) // s + 1
        if x is None:
            fwd = lambda: _C.conv_fwd_u8(frames, w, b, k, k, s, s, 1 / 255.0, -0.5, 1)
        else:
            xin = x
            fwd = lambda: _C.conv_fwd_nhwc(xin, w, b, k, k, s, s, 1)
        y, _ = fwd()
        dy = torch.randn_like(y); dw = torch.empty_like(w); db = torch.empty_like(b)
        if x is None:
            bww = lambda: _C.conv_bwd_weight_u8(dy, y, 1, frames, k, k, s, s, 1 / 255.0, -0.5, dw, db)
        else:
            bww = lambda: _C.conv_bwd_weight_nhwc(dy, y, 1, xin, k, k, s, s, dw, db)
        fl = 2.0 * B * Ho * Ho * C * k * k * Co
        tf_, tw_ = timed(fwd, 100), timed(bww, 100)
        print(json.dumps(dict(kernel=name, M=B * Ho * Ho, K=C * k * k, N=Co, fwd_us=round(tf_, 1), bwd_w_us=round(tw_, 1),
                              fwd_tf=round(fl / tf_ * 1e-6, 1), bwd_w_tf=round(fl / tw_ * 1e-6, 1))), flush=True)
        x = y.view(B, Ho, Ho, Co)


def clk(M=4096, K=256, N=256):
    """TRL_LIB=<clk build>: phase stamps of workgroups 0, 32, .. 224 of one launch of each kernel."""
    import ctypes as C
    import numpy as np
    dev = torch.device("cuda:0")
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    y = _C.linear_fwd(x, w, b, 1); dy = torch.randn(M, N, device=dev)
    ws = torch.empty(_C.lib().trl_linear_bwd_weight_workspace(M, K, N), device=dev)
    names = ["start", "fetch0 issued", "panel0 in LDS", "mfma0 done", "panel1 in LDS", "mfma1 done", "epilogue done"]
    for tag, fn in (("fwd", lambda: _C.linear_fwd(x, w, b, 1)), ("bwd_in", lambda: _C.linear_bwd_input(dy, y, 1, w)),
                    ("bwd_w", lambda: _C.linear_bwd_weight(dy, y, 1, x, workspace=ws))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); e.record()
        torch.cuda.synchronize()
        buf = np.zeros(128, dtype=np.int64)
        _C.lib().trl_dbg_gemm_clk.argtypes = [C.c_void_p]
        _C.lib().trl_dbg_gemm_clk(buf.ctypes.data)
        buf = buf.reshape(8, 8, 2)
        t0 = buf[:, 0, 1].min()
        print("%s (%d, %d, %d): event time %.1f us" % (tag, M, K, N, a.elapsed_time(e) * 1e3))
        for wg in range(8):
            cyc = buf[wg, :7, 0] - buf[wg, 0, 0]
            rt = (buf[wg, :7, 1] - t0) * 0.01
            mhz = (buf[wg, 6, 0] - buf[wg, 0, 0]) / max(1e-9, (buf[wg, 6, 1] - buf[wg, 0, 1]) * 0.01)
            print("  wg %3d: start +%.2f us; cycles %s; us %s; clock %.0f MHz" % (
                32 * wg, rt[0], " ".join("%d" % c for c in cyc[1:]), " ".join("%.2f" % t for t in (rt[1:] - rt[0])), mhz))


def clk_conv():
    """TRL_LIB=<clk build>: phase stamps of the implicit first conv layer at cfg 5 (512 x 4 x 84 x 84 uint8, 8x8 s4 -> 16)."""
    import ctypes as C
    import numpy as np
    dev = torch.device("cuda:0")
    frames = torch.randint(0, 256, (512, 4, 84, 84), dtype=torch.uint8, device=dev)
    w = torch.randn(16, 256, device=dev) * 0.05; b = torch.randn(16, device=dev)
    y, _ = _C.conv_fwd_u8(frames, w, b, 8, 8, 4, 4, 1 / 255.0, -0.5, 1)
    dy = torch.randn_like(y); dw = torch.empty_like(w); db = torch.empty_like(b)
    names = ["start", "fetch0 issued", "panel0 in LDS", "mfma0 done", "panel1 in LDS", "mfma1 done", "epilogue done"]
    for tag, fn in (("conv1 fwd", lambda: _C.conv_fwd_u8(frames, w, b, 8, 8, 4, 4, 1 / 255.0, -0.5, 1)),
                    ("conv1 bwd_w", lambda: _C.conv_bwd_weight_u8(dy, y, 1, frames, 8, 8, 4, 4, 1 / 255.0, -0.5, dw, db))):
        t_us = timed(fn, 50)
        torch.cuda.synchronize()
        buf = np.zeros(128, dtype=np.int64)
        _C.lib().trl_dbg_gemm_clk.argtypes = [C.c_void_p]
        _C.lib().trl_dbg_gemm_clk(buf.ctypes.data)
        buf = buf.reshape(8, 8, 2)
        print("%s: %.1f us per call" % (tag, t_us))
        for wg in (0, 3):
            cyc = buf[wg, :7, 0] - buf[wg, 0, 0]
            print("  wg %3d: cycles %s" % (32 * wg, " ".join("%d" % c for c in cyc[1:])))


def big():
    """Large dense products (TRL_GEMM_TILE=64 / 128 pins the workgroup tile; TRL_LIB + TRL_LIB_LAX=1: an older build):
    the three dense-layer GEMMs at 4096^3 / 8192 x 4096 x 4096 / 2048^3, and the vendor library's SGEMM as the yardstick."""
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for M, K, N in ((4096, 4096, 4096), (2048, 2048, 2048), (8192, 4096, 4096), (4096, 1024, 1024)):
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.02
        y = _C.linear_fwd(x, w, None, _C.ACT_NONE)
        ref = x @ w.t()
        err = float((y - ref).abs().max() / ref.abs().max())
        dy = torch.randn(M, N, device=dev)
        ws = torch.empty(_C.lib().trl_linear_bwd_weight_workspace(M, K, N), device=dev)
        dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
        fl = 2.0 * M * K * N
        t_f = timed(lambda: _C.linear_fwd(x, w, None, _C.ACT_NONE), 10)
        t_i = timed(lambda: _C.linear_bwd_input(dy, y, 1, w), 10)
        t_w = timed(lambda: _C.linear_bwd_weight(dy, y, 1, x, dw=dw, db=db, workspace=ws), 10)
        t_l = timed(lambda: torch.mm(x, w.t()), 10)
        print(json.dumps(dict(M=M, K=K, N=N, tile=os.environ.get("TRL_GEMM_TILE", "auto"), lib=os.path.basename(_C.LIB_PATH),
                              fwd_tf=round(fl / t_f * 1e-6, 1), bwd_in_tf=round(fl / t_i * 1e-6, 1),
                              bwd_w_tf=round(fl / t_w * 1e-6, 1), torch_mm_tf=round(fl / t_l * 1e-6, 1), rel_err=err)), flush=True)


if __name__ == "__main__":
    if "--big" in sys.argv:
        big()
        sys.exit(0)
    if "--grouped" in sys.argv:
        grouped()
        sys.exit(0)
    if "--clk-conv" in sys.argv:
        clk_conv()
        sys.exit(0)
    if "--clk" in sys.argv:
        clk()
        sys.exit(0)
    main()
