#!/usr/bin/env python
"""Scratch: per-launch time series of the 64^2 self-attention at 8 and 16 frames (is the 16-frame launch slower per block?)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mgld_vsr_amd import hip
from tools.attn_sp_check import run, LOG2E
DEV = "cuda"
hip.lib()
H, N = int(os.environ.get("PROBE_H", 5)), int(os.environ.get("PROBE_N", 4096))
C_ = H * 64
for B in (8, 16, 8, 16, 24, 32):
    qkv = torch.randn(B * N, 3 * C_, device=DEV)
    qkv[:, :C_] *= 64 ** -0.5 * LOG2E
    qkv = qkv.half()
    ev = [hip.Event() for _ in range(13)]
    run(qkv, B, H, N)
    ev[0].record()
    for i in range(12):
        run(qkv, B, H, N)
        ev[i + 1].record()
    ev[-1].sync()
    ts = [1e3 * ev[i].elapsed_ms(ev[i + 1]) for i in range(12)]
    print(f"B={B:2d}: " + " ".join(f"{t:6.1f}" for t in ts) + f"   us/frame {sum(ts) / 12 / B:6.2f}", flush=True)
    if B == 16:   # the same work as two 8-frame launches
        ev[0].record()
        for i in range(6):
            hw = B * N // 2
            run(qkv[:hw], 8, H, N); run(qkv[hw:], 8, H, N)
        ev[1].record(); ev[1].sync()
        print(f"   as 2 x 8 frames: {1e3 * ev[0].elapsed_ms(ev[1]) / 6:6.1f} us per pair")
