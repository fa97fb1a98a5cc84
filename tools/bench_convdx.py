"""Phase breakdown of the implicit conv input-gradient kernel (k_conv_dx.hip) at cfg 5's two geometries: HIP-event time
of trl_conv_bwd_input_nhwc_f32 with phases switched off (needs a -DTRL_EXP_DX build: python torchrl_amd/build.py --exp dx
-DTRL_EXP_DX; TRL_LIB=torchrl_amd/lib/libtrl_hip_dx.so python tools/bench_convdx.py).  Development aid."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from torchrl_amd import _C  # noqa: E402

DEV = torch.device("cuda:0")
GEOMS = {"conv2": (512, 16, 20, 20, 4, 4, 2, 2, 32), "conv3": (512, 32, 9, 9, 3, 3, 1, 1, 64)}


def timed(fn, reps=20):
    """GPU time per call: `reps` calls captured in a HIP graph (no host launch cost in between), replayed 5 times."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(reps):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        graph.replay()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / (5 * reps)


def main():
    for name, (B, C, H, W, kh, kw, sh, sw, Co) in GEOMS.items():
        Ho, Wo = (H - kh) // sh + 1, (W - kw) // sw + 1
        dy, y = torch.randn(B * Ho * Wo, Co, device=DEV), torch.randn(B * Ho * Wo, Co, device=DEV)
        w = torch.randn(Co, C * kh * kw, device=DEV)
        for label, dbg, gate, tpw in (("full", 0, y, None), ("no gate", 0, None, None), ("no staging", 1, y, None),
                                      ("no A loads", 2, y, None), ("no stores", 4, y, None), ("no mfma", 8, y, None),
                                      ("nothing", 15, y, None), ("tpw 2", 0, y, 2), ("tpw 4", 0, y, 4), ("tpw 8", 0, y, 8)):
            os.environ["TRL_DX_DBG"] = str(dbg)
            if tpw is None:
                os.environ.pop("TRL_DX_TPW", None)
            else:
                os.environ["TRL_DX_TPW"] = str(tpw)
            us = timed(lambda: _C.conv_bwd_input_nhwc(dy, gate, _C.ACT_TANH, w, B, C, H, W, kh, kw, sh, sw))
            print("%s %-11s %7.1f us (prep launch included)" % (name, label, us), flush=True)
        old = timed(lambda: _C.col2im(_C.linear_bwd_input(dy, y, _C.ACT_TANH, w), B, C, H, W, kh, kw, sh, sw))
        print("%s cols GEMM + col2im %7.1f us" % (name, old), flush=True)


if __name__ == "__main__":
    main()
