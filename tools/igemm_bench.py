#!/usr/bin/env python
"""Kernel-tuning microbench for the implicit-GEMM: times representative problems of the hot path (taken from the per-problem
table bench.py dumps) in isolation with hipEvents.  `MGLD_IGEMM_FORCE=<BM*1000+BN>` overrides the launcher's tile choice.
Scratch tool — not part of the product path or the test suite."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mgld_vsr_amd import hip  # noqa: E402

# (name, mode, M, N, K, Cin, H(in=out), act, weight = launches per 8x512^2 segment)
SHAPES = [
    ("conv64 320->320", 1, 32768, 320, 2880, 320, 64, 0, 350),
    ("conv64 640->320", 1, 32768, 320, 5760, 640, 64, 0, 100),
    ("conv64 128->640 (spade gb)", 1, 32768, 640, 1152, 128, 64, 0, 250),
    ("lin64 320->320", 0, 32768, 320, 320, 0, 0, 0, 1050),
    ("geglu64 320->2560", 0, 32768, 2560, 320, 0, 0, 4, 250),
    ("lin64 1280->320", 0, 32768, 320, 1280, 0, 0, 0, 250),
    ("conv32 640->640", 1, 8192, 640, 5760, 640, 32, 0, 300),
    ("geglu32 640->5120", 0, 8192, 5120, 640, 0, 0, 4, 250),
    ("lin32 640->640", 0, 8192, 640, 640, 0, 0, 0, 1050),
    ("conv16 1280->1280", 1, 2048, 1280, 11520, 1280, 16, 0, 300),
    ("lin16 1280->1280", 0, 2048, 1280, 1280, 0, 0, 0, 1050),
    ("conv8 1280->1280", 1, 512, 1280, 11520, 1280, 8, 0, 550),
    ("lin16 5120->1280", 0, 2048, 1280, 5120, 0, 0, 0, 250),
    ("lin16 2560->1280", 0, 2048, 1280, 2560, 0, 0, 0, 100),
    ("conv8 512->512", 1, 512, 512, 4608, 512, 8, 0, 400),
    ("conv32 256->256", 1, 8192, 256, 2304, 256, 32, 0, 300),
    ("conv64 256->128", 1, 32768, 128, 2304, 256, 64, 0, 250),
    ("conv16 2560->1280", 1, 2048, 1280, 23040, 2560, 16, 0, 100),
    ("vae conv256 256->256", 1, 524288, 256, 2304, 256, 256, 0, 13),
    ("vae conv512 128->128", 1, 2097152, 128, 1152, 128, 512, 0, 13),
]


def main():
    hip.lib()
    hip.ensure_workspace()
    dev = "cuda"
    e0, e1 = hip.Event(), hip.Event()
    tot_ms = 0.0
    rows = []
    only = os.environ.get("MGLD_BENCH_ONLY")            # e.g. "0,12": indices into SHAPES (for PMC passes)
    iters = int(os.environ.get("MGLD_BENCH_ITERS", "20"))
    shapes = [SHAPES[int(i)] for i in only.split(",")] if only else SHAPES
    for name, mode, M, N, K, Cin, H, act, weight in shapes:
        if mode == 1:
            frames = M // (H * H)
            a = torch.randn(frames * H * H, Cin, device=dev).half()
        else:
            a = torch.randn(M, K, device=dev).half()
        w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
        out = torch.empty(M, N // 2 if act == 4 else N, dtype=torch.half, device=dev)
        bias = torch.randn(N, device=dev)

        tiled = (mode == 1 and os.environ.get("MGLD_BENCH_TILED", "1") == "1" and H <= 64 and hip.conv3p_applies(M // (H * H), Cin, N, H, H))
        if tiled:   # weight values are random anyway: any [.., 32] tensor of the right size is a valid tiled layout
            w = (torch.randn((N + 63) // 64 * 64 * 9 * Cin // 32, 32, device=dev) * K ** -0.5).half()

        def launch():
            if tiled:
                hip.igemm(a, w, out, mode=1, bias=bias, conv=(Cin, H, H, H, H, 1, 1, 1, 0), tap_inner=2, N=N, K=K)
            elif mode == 1:
                hip.igemm(a, w, out, mode=1, bias=bias, conv=(Cin, H, H, H, H, 1, 1, 1, 0),
                          tap_inner=int(os.environ.get("MGLD_TAP_INNER", "1")) if Cin % 64 == 0 else 0)
            else:
                hip.igemm(a, w, out, bias=bias, act=act)
        for _ in range(3):
            launch()
        e0.record()
        for _ in range(iters):
            launch()
        e1.record()
        e1.sync()
        us = 1e3 * e0.elapsed_ms(e1) / iters
        tf = 2.0 * M * N * K / (us * 1e-6) / 1e12
        tot_ms += us * weight / 1e3
        rows.append((name, us, tf))
        print(f"{name:32s} M={M:8d} N={N:5d} K={K:6d}  {us:9.2f} us  {tf:7.1f} TF/s")
    print(f"weighted total: {tot_ms:.1f} ms   (MGLD_IGEMM_FORCE={os.environ.get('MGLD_IGEMM_FORCE', '-')})")


if __name__ == "__main__":
    main()
