"""A/B of the two forms of the implicit conv input gradient (k_conv_dx.hip) at cfg 5's geometries: the class form (dZ re-read
through L1 per tap) against the image-tile form (gated dZ of whole images staged in LDS once).  Outputs must be bit-identical;
times are HIP-graph replays (tools/bench_convdx.py::timed).  TRL_DX_IMG / TRL_DX_WAVES / TRL_DX_CLASS_FORM are read per launch."""
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
from torchrl_amd import _C  # noqa: E402
from bench_convdx import DEV, GEOMS, timed  # noqa: E402


def main():
    out = {}
    for name, (B, C, H, W, kh, kw, sh, sw, Co) in GEOMS.items():
        Ho, Wo = (H - kh) // sh + 1, (W - kw) // sw + 1
        gen = torch.Generator(device=DEV).manual_seed(1)
        dy = torch.randn(B * Ho * Wo, Co, device=DEV, generator=gen)
        y = torch.tanh(torch.randn(B * Ho * Wo, Co, device=DEV, generator=gen))
        xg = torch.tanh(torch.randn(B * H * W, C, device=DEV, generator=gen))
        w = torch.randn(Co, C * kh * kw, device=DEV, generator=gen) * 0.05
        call = lambda: _C.conv_bwd_input_nhwc(dy, y, _C.ACT_TANH, w, B, C, H, W, kh, kw, sh, sw, x_gate=xg, x_gate_act=_C.ACT_TANH)
        for k in ("TRL_DX_IMG", "TRL_DX_WAVES"):
            os.environ.pop(k, None)
        os.environ["TRL_DX_CLASS_FORM"] = "1"
        ref = call().clone()
        res = {"class_form_us": timed(call)}
        os.environ["TRL_DX_CLASS_FORM"] = "0"
        got = call().clone()
        res["default_us"] = timed(call)
        res["bit_identical"] = bool(torch.equal(ref, got))
        res["max_abs_diff"] = float((ref - got).abs().max())
        for img in (1, 2, 4):
            for waves in (4, 8):
                os.environ["TRL_DX_IMG"], os.environ["TRL_DX_WAVES"] = str(img), str(waves)
                ok = bool(torch.equal(call(), ref))
                res["img%d_waves%d_us" % (img, waves)] = round(timed(call), 2)
                res["img%d_waves%d_equal" % (img, waves)] = ok
        # small / ragged batches (partial last workgroup, one image)
        for b in (1, 3, 7):
            for k in ("TRL_DX_IMG", "TRL_DX_WAVES"):
                os.environ.pop(k, None)
            os.environ["TRL_DX_IMG"] = "2"
            sub = lambda: _C.conv_bwd_input_nhwc(dy[:b * Ho * Wo], y[:b * Ho * Wo], _C.ACT_TANH, w, b, C, H, W, kh, kw, sh, sw,
                                                 x_gate=xg[:b * H * W], x_gate_act=_C.ACT_TANH)
            os.environ["TRL_DX_CLASS_FORM"] = "1"
            r = sub().clone()
            os.environ["TRL_DX_CLASS_FORM"] = "0"
            res["B%d_equal" % b] = bool(torch.equal(sub(), r))
        out[name] = res
        print(name, json.dumps(res), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
