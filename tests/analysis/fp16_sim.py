#!/usr/bin/env python
"""What-if study of the fp16 storage floor (CPU, no GPU): the oracle networks with fp16 rounding injected at chosen points, against
the fp32 golden outputs.  R(x) = x.half().float().  Modes: which tensors are rounded
   in   - activations entering a contraction (conv / linear / conv1d / conv3d / attention matmuls): what the MFMA sees in any case
   w    - weights of the contractions
   out  - outputs of contractions and of the normalisations as they are stored (fp16 tensors between kernels)
Analysis script behind the floor table of DESIGN.md section 5 (test infrastructure: it drives the oracle; not collected by pytest)."""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from configs import STRUCT_SMALL, T, UNET_SMALL, VAE_DD_SMALL  # noqa: E402
from mgld_vsr_amd import synth  # noqa: E402
from oracle import nets  # noqa: E402

R = lambda t: t.half().float()
MODE = {"in": False, "w": False, "out": False, "norm_out": False, "out_lin": None, "out_conv": None}
_orig = {k: getattr(F, k) for k in ("conv2d", "linear", "conv1d", "conv3d", "group_norm", "layer_norm")}
_einsum = torch.einsum


def _contract(name):
    f = _orig[name]

    def g(x, w, b=None, *a, **k):
        x = R(x) if MODE["in"] else x
        w = R(w) if MODE["w"] else w
        y = f(x, w, b, *a, **k)
        o = MODE["out"]
        if name == "linear" and MODE["out_lin"] is not None:
            o = MODE["out_lin"]
        if name != "linear" and MODE["out_conv"] is not None:
            o = MODE["out_conv"]
        return R(y) if o else y
    return g


def _norm(name):
    f = _orig[name]

    def g(x, *a, **k):
        y = f(x, *a, **k)
        return R(y) if MODE["norm_out"] else y
    return g


def _ein(eq, a, b):
    if MODE["in"]:
        a, b = R(a), R(b)
    y = _einsum(eq, a, b)
    return y


def install():
    for k in ("conv2d", "linear", "conv1d", "conv3d"):
        setattr(F, k, _contract(k))
    for k in ("group_norm", "layer_norm"):
        setattr(F, k, _norm(k))
    torch.einsum = _ein


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def G(name):
    d = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    return {k: (torch.from_numpy(d[k]) if d[k].dtype.kind in "fiu" else d[k]) for k in d.files}


def main():
    install()
    torch.set_num_threads(8)
    g = G("g_unet")
    usd = synth.synth_state_dict(json.loads(str(g["unet_params"])), "unet")
    sc = {k[3:]: v for k, v in g.items() if k.startswith("sc_")}
    gv = G("g_vae")
    vsd = synth.synth_state_dict(json.loads(str(gv["vae_params"])), "vae")
    cases = [("fp32", {}), ("in", dict(in_=1)), ("w", dict(w=1)), ("in+w", dict(in_=1, w=1)), ("in+w+norm_out", dict(in_=1, w=1, norm_out=1)),
             ("in+w+out (all stored fp16)", dict(in_=1, w=1, out=1, norm_out=1)), ("out only", dict(out=1, norm_out=1)),
             ("all, linear outputs fp32", dict(in_=1, w=1, out=1, norm_out=1, out_lin=0)),
             ("all, conv outputs fp32", dict(in_=1, w=1, out=1, norm_out=1, out_conv=0))]
    for name, m in cases:
        MODE.update({"in": bool(m.get("in_")), "w": bool(m.get("w")), "out": bool(m.get("out")), "norm_out": bool(m.get("norm_out")),
                     "out_lin": m.get("out_lin"), "out_conv": m.get("out_conv")})
        with torch.no_grad():
            eps = nets.unet_forward(usd, UNET_SMALL, g["x"], g["t"], g["ctx"], sc)
            dec = nets.vae_decode(vsd, VAE_DD_SMALL, gv["z"], [gv["fea0"], gv["fea1"]], fusion_w=1.0)
            dec05 = nets.vae_decode(vsd, VAE_DD_SMALL, gv["z"], [gv["fea0"], gv["fea1"]], fusion_w=0.5)
        print(f"{name:32s} unet {rel(eps, g['eps']):.2e}   vae dec {rel(dec, gv['dec']):.2e}   dec_w05 {rel(dec05, gv['dec_w05']):.2e}", flush=True)


if __name__ == "__main__":
    main()
