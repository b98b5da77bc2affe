#!/usr/bin/env python
"""Scratch: the SP attention kernel at one vs two blocks per CU (LDS request as the occupancy knob)."""
import os, sys, torch
os.environ["MGLD_DEBUG_DYNENV"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mgld_vsr_amd import hip
from tools.attn_sp_check import run, LOG2E
hip.lib()
B, H, N = 16, 5, 4096
C_ = H * 64
qkv = torch.randn(B * N, 3 * C_, device="cuda"); qkv[:, :C_] *= 64 ** -0.5 * LOG2E; qkv = qkv.half()
e0, e1 = hip.Event(), hip.Event()
for sp in sys.argv[1:] or ["2"]:
    os.environ["MGLD_ATTN_SP"] = sp
    for lds in (65536, 100000, 65536, 100000):
        os.environ["MGLD_ATTN_SP_LDS"] = str(lds)
        run(qkv, B, H, N)
        e0.record()
        for _ in range(5):
            run(qkv, B, H, N)
        e1.record(); e1.sync()
        us = 1e3 * e0.elapsed_ms(e1) / 5
        print(f"sp={sp} lds={lds}: {us:8.1f} us", flush=True)
