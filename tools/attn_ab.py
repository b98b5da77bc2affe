#!/usr/bin/env python
"""Scratch: A/B of attention kernel variants INSIDE one process (MGLD_DEBUG_DYNENV: the library re-reads MGLD_ATTN_SP at every launch), variants
interleaved round-robin so clock / thermal drift hits all of them alike.  usage: attn_ab.py <variants...>   (0 = the round-5 kernel)"""
import os, sys, torch
os.environ["MGLD_DEBUG_DYNENV"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mgld_vsr_amd import hip
from tools.attn_sp_check import run, LOG2E
DEV = "cuda"
hip.lib()
variants = sys.argv[1:] or ["0", "1", "2"]
shapes = [("64^2 x16", 16, 5, 4096), ("32^2 x16", 16, 10, 1024), ("16^2 x16", 16, 20, 256), ("64^2 x8", 8, 5, 4096)]
if os.environ.get("AB_SHAPES"):
    shapes = [s for s in shapes if s[0] in os.environ["AB_SHAPES"].split(",")]
for name, B, H, N in shapes:
    C_ = H * 64
    qkv = torch.randn(B * N, 3 * C_, device=DEV)
    qkv[:, :C_] *= 64 ** -0.5 * LOG2E
    qkv = qkv.half()
    tot = {v: 0.0 for v in variants}
    best = {v: 1e9 for v in variants}
    e0, e1 = hip.Event(), hip.Event()
    R = 8
    for r in range(R + 1):
        for v in variants:
            os.environ["MGLD_ATTN_SP"] = v
            run(qkv, B, H, N)
            e0.record()
            for _ in range(5):
                run(qkv, B, H, N)
            e1.record(); e1.sync()
            us = 1e3 * e0.elapsed_ms(e1) / 5
            if r > 0:
                tot[v] += us; best[v] = min(best[v], us)
    fl = 4.0 * B * H * N * N * 64
    print(name + ": " + "  ".join(f"[{v}] {tot[v] / R:7.1f} us (min {best[v]:7.1f}) {fl / (tot[v] / R * 1e-6) / 2.5e15:.3f}" for v in variants), flush=True)
