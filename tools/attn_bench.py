#!/usr/bin/env python
"""Kernel-tuning microbench for the flash-attention kernel on the shapes of the 8x512^2 segment (hipEvent timing).
Scratch tool — not part of the product path or the test suite."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mgld_vsr_amd import hip  # noqa: E402

# (name, frames, heads, Nq, Nkv, D, launches per segment)
SHAPES = [
    ("self 64^2  h5", 8, 5, 4096, 4096, 64, 250),
    ("self 32^2 h10", 8, 10, 1024, 1024, 64, 250),
    ("self 16^2 h20", 8, 20, 256, 256, 64, 250),
    ("cross 64^2 h5", 8, 5, 4096, 77, 64, 250),
    ("cross 32^2 h10", 8, 10, 1024, 77, 64, 250),
]


def main():
    hip.lib()
    dev = "cuda"
    e0, e1 = hip.Event(), hip.Event()
    tot = 0.0
    for name, B, H, Nq, Nkv, D, weight in SHAPES:
        C = H * D
        Np = (Nkv + 7) // 8 * 8
        q = (torch.randn(B * Nq, C, device=dev) * (D ** -0.5 * 1.4426950408889634 if (Nkv == Nq and os.environ.get("ATTN_BENCH_PS", "1") != "0") else 1.0)).half()
        k = torch.randn(B * Nkv, C, device=dev).half()
        vt = torch.randn(B * C, Np, device=dev).half()
        v = torch.randn(B * Nkv, C, device=dev).half()
        o = torch.empty_like(q)
        vrm = Nkv == Nq          # self-attention: V row-major out of the fused q|k|v projection; cross-attention: cached V^T

        ps = os.environ.get("ATTN_BENCH_PS", "1") != "0"      # self-attention as the UNet launches it since round 5: pre-scaled queries

        def launch():
            if vrm:
                hip.attention(q, k, v, o, batch=B, heads=H, Nq=Nq, Nkv=Nkv, head_dim=D, q_strides=(Nq * C, C, D),
                              k_strides=(Nkv * C, C, D), vt_strides=(Nkv * C, C, D), o_strides=(Nq * C, C, D),
                              scale=(1.0 / 1.4426950408889634) if ps else D ** -0.5, v_rowmajor=True)
            else:
                hip.attention(q, k, vt, o, batch=B, heads=H, Nq=Nq, Nkv=Nkv, head_dim=D, q_strides=(Nq * C, C, D),
                              k_strides=(Nkv * C, C, D), vt_strides=(C * Np, D * Np, Np), o_strides=(Nq * C, C, D), scale=D ** -0.5)
        for _ in range(3):
            launch()
        e0.record()
        for _ in range(20):
            launch()
        e1.record()
        e1.sync()
        us = 1e3 * e0.elapsed_ms(e1) / 20
        tf = 4.0 * B * H * Nq * Nkv * D / (us * 1e-6) / 1e12
        tot += us * weight / 1e3
        print(f"{name:16s} B={B} H={H:2d} Nq={Nq:5d} Nkv={Nkv:5d}  {us:9.2f} us  {tf:7.1f} TF/s")
    print(f"weighted total: {tot:.1f} ms")


if __name__ == "__main__":
    main()
