#!/usr/bin/env python
"""Kernel-tuning microbench for the GroupNorm kernels (stats / apply / SPADE apply) on the shapes of the 8x512^2 segment.
Scratch tool — not part of the product path or the test suite."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mgld_vsr_amd import hip  # noqa: E402

# (frames, rows per frame, C)
SHAPES = [(16, 4096, 320), (16, 4096, 640), (16, 4096, 960), (16, 1024, 640), (16, 1024, 1280), (16, 256, 1280), (8, 4096, 320), (8, 4096, 640), (8, 4096, 960), (8, 1024, 640), (8, 1024, 1280), (8, 1024, 1920),
          (8, 256, 1280), (8, 256, 2560), (8, 64, 1280), (8, 64, 2560), (8, 16384, 512), (8, 65536, 256), (8, 262144, 128)]


def timeit(fn, e0, e1, iters=20):
    for _ in range(3):
        fn()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.sync()
    return 1e3 * e0.elapsed_ms(e1) / iters


def main():
    hip.lib()
    dev = "cuda"
    e0, e1 = hip.Event(), hip.Event()
    for frames, rows, C in SHAPES:
        x = torch.randn(frames * rows, C, device=dev).half()
        y = torch.empty_like(x)
        gs = torch.empty(frames, hip.gn_chunks(rows), 32, 2, dtype=torch.float64, device=dev)
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        mb = x.numel() * 2 / 1e6
        t_s = timeit(lambda: hip.gn_stats(x, frames, rows, 32, gs), e0, e1)
        t_a = timeit(lambda: hip.gn_apply(x, gs, 1e-5, g, b, y, frames, rows, 32, True), e0, e1)
        line = f"frames={frames} rows={rows:6d} C={C:5d} ({mb:7.1f} MB)  stats {t_s:8.2f} us {mb / t_s:6.2f} TB/s   apply {t_a:8.2f} us {2 * mb / t_a:6.2f} TB/s"
        if not hip.gn_fused_applies(rows, C, 32):
            xl = torch.randn(frames * rows, C, device=dev).half()
            t_l = timeit(lambda: hip.gn_apply(x, gs, 1e-5, g, b, y, frames, rows, 32, True, x_lo=xl), e0, e1)
            line += f"   apply_lo {t_l:8.2f} us {3 * mb / t_l:6.2f} TB/s"
        if rows <= 4096:
            gb = torch.randn(frames * rows, 2 * C, device=dev).half()
            sk = torch.randn(frames * rows, C, device=dev).half()
            t_p = timeit(lambda: hip.spade_apply(x, gs, 1e-5, g, b, gb, sk, y, frames, rows, 32), e0, e1)
            line += f"   spade {t_p:8.2f} us {5 * mb / t_p:6.2f} TB/s"
        if hip.gn_fused_applies(rows, C, 32):
            t_f = timeit(lambda: hip.gn_fused(x, 1e-5, g, b, y, frames, rows, 32, 1), e0, e1)
            line += f"   fused {t_f:8.2f} us {2 * mb / t_f:6.2f} TB/s"
        print(line)


if __name__ == "__main__":
    main()
