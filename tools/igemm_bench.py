#!/usr/bin/env python
"""Kernel-tuning microbench for the GEMM family: times representative problems of the hot path (shapes + launch counts of one 8 x 512^2
segment, from the per-problem table bench.py dumps) in isolation with hipEvents, for each requested kernel variant / tile order, interleaved
in ONE process (within-probe A/B).  Operand buffers are rotated over > 256 MiB so that repeated launches do not find their inputs in the
Infinity Cache.  Scratch tool — not part of the product path or the test suite.

    python tools/igemm_bench.py [conv|lin|vae|all] [--variants 0,1,2] [--rounds 3]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mgld_vsr_amd import hip  # noqa: E402

# (name, mode, M, N, K, Cin, H(in), W(in), act, up2, launches per segment)
CONV = [
    ("conv64 320->320", 1, 32768, 320, 2880, 320, 64, 64, 0, 0, 350),
    ("conv64 640->320", 1, 32768, 320, 5760, 640, 64, 64, 0, 0, 100),
    ("conv64 128->640 spade", 1, 32768, 640, 1152, 128, 64, 64, 0, 0, 250),
    ("conv64 256->128", 1, 32768, 128, 2304, 256, 64, 64, 0, 0, 250),
    ("conv32 640->640", 1, 8192, 640, 5760, 640, 32, 32, 0, 0, 300),
    ("conv32 256->256", 1, 8192, 256, 2304, 256, 32, 32, 0, 0, 300),
    ("conv16 1280->1280", 1, 2048, 1280, 11520, 1280, 16, 16, 0, 0, 300),
    ("conv16 2560->1280", 1, 2048, 1280, 23040, 2560, 16, 16, 0, 0, 100),
    ("conv8 1280->1280", 1, 512, 1280, 11520, 1280, 8, 8, 0, 0, 550),
    ("conv8 2560->1280", 1, 512, 1280, 23040, 2560, 8, 8, 0, 0, 150),
    ("up 32->64 640", 1, 32768, 640, 5760, 640, 32, 32, 0, 1, 50),
    ("up 16->32 1280", 1, 8192, 1280, 11520, 1280, 16, 16, 0, 1, 50),
]
VAE = [
    ("vae64 512->512", 1, 32768, 512, 4608, 512, 64, 64, 0, 0, 30),
    ("vae128 512->512", 1, 131072, 512, 4608, 512, 128, 128, 0, 0, 20),
    ("vae256 256->256", 1, 524288, 256, 2304, 256, 256, 256, 0, 0, 25),
    ("vae512 128->128", 1, 2097152, 128, 1152, 128, 512, 512, 0, 0, 25),
    ("vae up 64->128 512", 1, 131072, 512, 4608, 512, 64, 64, 0, 1, 1),
    ("vae up 128->256 512", 1, 524288, 512, 4608, 512, 128, 128, 0, 1, 1),
    ("vae up 256->512 256", 1, 2097152, 256, 2304, 256, 256, 256, 0, 1, 1),
]
LIN = [
    ("lin64 320->320", 0, 32768, 320, 320, 0, 0, 0, 0, 0, 1250),
    ("lin64 320->960 qkv", 0, 32768, 960, 320, 0, 0, 0, 0, 0, 250),
    ("geglu64 320->2560", 0, 32768, 2560, 320, 0, 0, 0, 4, 0, 250),
    ("lin64 1280->320", 0, 32768, 320, 1280, 0, 0, 0, 0, 0, 250),
    ("lin32 640->640", 0, 8192, 640, 640, 0, 0, 0, 0, 0, 1250),
    ("lin32 640->1920 qkv", 0, 8192, 1920, 640, 0, 0, 0, 0, 0, 250),
    ("geglu32 640->5120", 0, 8192, 5120, 640, 0, 0, 0, 4, 0, 250),
    ("lin32 2560->640", 0, 8192, 640, 2560, 0, 0, 0, 0, 0, 250),
    ("lin16 1280->1280", 0, 2048, 1280, 1280, 0, 0, 0, 0, 0, 1250),
    ("lin16 1280->3840 qkv", 0, 2048, 3840, 1280, 0, 0, 0, 0, 0, 250),
    ("geglu16 1280->10240", 0, 2048, 10240, 1280, 0, 0, 0, 4, 0, 250),
    ("lin16 5120->1280", 0, 2048, 1280, 5120, 0, 0, 0, 0, 0, 250),
    ("lin8 1280->1280", 0, 512, 1280, 1280, 0, 0, 0, 0, 0, 300),
    ("vae attn 256->768", 0, 327680, 768, 256, 0, 0, 0, 0, 0, 10),
    ("vae attn 256->256", 0, 327680, 256, 256, 0, 0, 0, 0, 0, 10),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--variants", default="0", help="conv: comma list of `tune` values (0 = launcher's choice, id+1 forces conv3q variant id)")
    ap.add_argument("--nst", default="0", help="lin: comma list of `tune` values (0 = launcher's choice, d-1 forces a d-deep LDS ring)")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--json", default=None)
    ap.add_argument("--only", default=None, help="comma list of substrings: run only the shapes whose name contains one of them")
    args = ap.parse_args()
    hip.lib()
    hip.ensure_workspace()
    dev = "cuda"
    shapes = {"conv": CONV, "vae": VAE, "lin": LIN, "all": CONV + VAE + LIN}[args.what]
    if args.only:
        shapes = [sh for sh in shapes if any(o.replace("_", " ") in sh[0] for o in args.only.split(","))]
    variants = [int(v) for v in args.variants.split(",")]
    nsts = [int(v) for v in args.nst.split(",")]
    allv = sorted(set(variants) | set(nsts))
    e0, e1 = hip.Event(), hip.Event()
    out_rows = []
    totals = {v: 0.0 for v in allv}
    mscale = int(os.environ.get("MGLD_BENCH_MSCALE", "1"))      # 2: the row counts of two segments batched as clips (bench.py --clips 2)
    for name, mode, M, N, K, Cin, H, W, act, up2, weight in shapes:
        M *= mscale
        sc = 2 if up2 else 1
        frames = M // (H * W * sc * sc) if mode == 1 else 0
        a_rows = frames * H * W if mode == 1 else M
        a_cols = Cin if mode == 1 else K
        nbuf = max(2, min(24, int((320 << 20) / max(1, 2 * (a_rows * a_cols + M * N))) + 1))   # rotate over > 256 MiB
        A = [torch.randn(a_rows, a_cols, device=dev).half() for _ in range(nbuf)]
        O = [torch.empty(M, N // 2 if act == 4 else N, dtype=torch.half, device=dev) for _ in range(nbuf)]
        bias = torch.randn(N, device=dev)
        tiled = mode == 1 and hip.conv3p_applies(frames, Cin, N, H, W, bool(up2))
        if tiled:   # weight values are random anyway: any [., 32] tensor of the right size is a valid tiled layout
            w = (torch.randn((N + 63) // 64 * 64 * 9 * Cin // 32, 32, device=dev) * K ** -0.5).half()
        else:
            w = (torch.randn(N, K, device=dev) * K ** -0.5).half()

        def launch(i, tune):
            a, o = A[i % nbuf], O[i % nbuf]
            if mode == 1:
                hip.igemm(a, w, o, mode=1, bias=bias, conv=(Cin, H, W, sc * H, sc * W, 1, 1, 1, up2), tap_inner=2 if tiled else 0, N=N, K=K,
                          tune=tune)
            else:
                hip.igemm(a, w, o, bias=bias, act=act, tune=tune)
        best = {}
        names = {}
        for v in (variants if mode == 1 else nsts):      # which kernel instantiation each variant runs (the planner may refuse a forced one)
            hip.IGEMM_LOG = []
            launch(0, v)
            names[v] = hip.igemm_kernel_name(hip.IGEMM_LOG[-1])[0].replace("_kernel", "").replace(" ", "")
            hip.IGEMM_LOG = None
        for r in range(args.rounds):
            for v in (variants if mode == 1 else nsts):
                if up2 and v > 2 and v != 8:
                    continue
                launch(0, v)
                e0.record()
                for i in range(args.iters):
                    launch(i + 1, v)
                e1.record()
                e1.sync()
                us = 1e3 * e0.elapsed_ms(e1) / args.iters
                best[v] = min(best.get(v, 1e30), us)
        p = hip.MgldIGemm()
        line = f"{name:26s} M={M:8d} N={N:5d} K={K:6d} "
        for v in allv:
            if v in best:
                tf = 2.0 * M * N * K / (best[v] * 1e-6) / 1e12
                totals[v] += best[v] * weight / 1e3
                line += f"| v{v} {names.get(v, '')[:34]}: {best[v]:8.2f} us {tf:7.1f} TF "
                out_rows.append({"name": name, "variant": v, "kernel": names.get(v, ""), "us": round(best[v], 2), "tflops": round(tf, 1), "weight": weight})
        print(line, flush=True)
        del A, O
    print("weighted totals (ms/segment): " + "  ".join(f"v{v}: {t:.1f}" for v, t in totals.items()) +
          f"   MGLD_IGEMM_ORDER={os.environ.get('MGLD_IGEMM_ORDER', 'auto')} MGLD_CONV3Q={os.environ.get('MGLD_CONV3Q', '1')}")
    if args.json:
        with open(args.json, "w") as fh:
            json.dump(out_rows, fh, indent=0)


if __name__ == "__main__":
    main()
