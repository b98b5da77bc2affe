#!/usr/bin/env python
"""Scratch: launch ONE attention shape a few times (for rocprofv3 --pmc runs): attn_one.py B H N [iters]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mgld_vsr_amd import hip
from tools.attn_sp_check import run, LOG2E
hip.lib()
B, H, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
it = int(sys.argv[4]) if len(sys.argv) > 4 else 5
C_ = H * 64
qkv = torch.randn(B * N, 3 * C_, device="cuda"); qkv[:, :C_] *= 64 ** -0.5 * LOG2E; qkv = qkv.half()
for _ in range(it):
    run(qkv, B, H, N)
torch.cuda.synchronize()
