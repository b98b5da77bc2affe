#!/usr/bin/env python
"""Microbench of the video VAE's temporal Conv3d (3,1,1) launches (MODE_TCONV3) at the decoder's four levels: the implicit-GEMM kernel in both row orders
(tune 15 consecutive rows, tune 14 frame-interleaved) against the planner's choice (tune 0: the ping-pong kernel of pptconv.hip where it covers), with and without the weight-residual pass.  Reports us per
launch and the algorithmic HBM rate (input rows once + output rows once).  Scratch tool — not product, not a test.

    python tools/tconv_bench.py [--iters 5] [--rounds 3]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mgld_vsr_amd import hip  # noqa: E402

# (name, T, HW, C)
SHAPES = [("vae512 C128", 8, 512 * 512, 128), ("vae256 C256", 8, 256 * 256, 256), ("vae128 C512", 8, 128 * 128, 512),
          ("vae64 C512", 8, 64 * 64, 512), ("vae1024 C128 T4", 4, 1024 * 1024, 128), ("unet8 C1280", 8, 64, 1280)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    hip.lib()
    hip.ensure_workspace()
    e0, e1 = hip.Event(), hip.Event()
    for name, T, HW, C in SHAPES:
        M = T * HW
        nbuf = max(2, min(8, int((640 << 20) / (4 * M * C)) + 1))
        A = [torch.randn(M, C, device="cuda").half() for _ in range(nbuf)]
        O = [torch.empty(M, C, dtype=torch.half, device="cuda") for _ in range(nbuf)]
        w = (torch.randn(C, 3 * C, device="cuda") * (3 * C) ** -0.5).half()
        w2 = (torch.randn(C, 3 * C, device="cuda")).half()
        bias = torch.randn(C, device="cuda")
        line = f"{name:18s} M={M:8d} "
        for two in (False, True):
            for tune in (15, 14, 0):   # consecutive rows, frame-interleaved rows (implicit-GEMM kernel), the planner (ping-pong kernel where covered)
                best = 1e30
                for _ in range(args.rounds):
                    def launch(i):
                        hip.igemm(A[i % nbuf], w, O[i % nbuf], mode=hip.MODE_TCONV3, bias=bias, resid=A[i % nbuf], alpha=0.6, beta=0.4,
                                  tconv=(C, T, HW), tune=tune, w2=w2 if two else None)
                    launch(0)
                    e0.record()
                    for i in range(args.iters):
                        launch(i + 1)
                    e1.record()
                    e1.sync()
                    best = min(best, 1e3 * e0.elapsed_ms(e1) / args.iters)
                gbs = 4.0 * M * C / (best * 1e-6) / 1e9
                line += f"| {'w2 ' if two else ''}{ {15: 'rows', 14: 'frames', 0: 'plan'}[tune] }: {best:8.1f} us {gbs:7.0f} GB/s "
        print(line, flush=True)
        del A, O


if __name__ == "__main__":
    main()
