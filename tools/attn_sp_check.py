#!/usr/bin/env python
"""Scratch: correctness + timing of ONE variant of the d = 64 self-attention kernel (env MGLD_ATTN_SP / MGLD_ATTN_PS / ... select it; the
kernel library reads them once per process, so tools/attn_sp_sweep.sh runs this script once per variant).  Not part of the product path."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mgld_vsr_amd import hip  # noqa: E402

DEV = "cuda"
LOG2E = 1.4426950408889634


def run(qkv, B, H, N, D=64):
    C_ = H * D
    o = torch.empty(B * N, C_, dtype=torch.half, device=DEV)
    st = (N * 3 * C_, 3 * C_, D)
    hip.attention(qkv, qkv[:, C_:], qkv[:, 2 * C_:], o, batch=B, heads=H, Nq=N, Nkv=N, head_dim=D, q_strides=st, k_strides=st, vt_strides=st,
                  o_strides=(N * C_, C_, D), scale=1.0 / LOG2E, v_rowmajor=True)
    return o


def check(B, H, N, adversarial=True, seed=78):
    D = 64
    C_ = H * D
    f = D ** -0.5 * LOG2E
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B * N, C_, generator=g) * 0.8
    k = torch.randn(B * N, C_, generator=g) * 0.8
    v = torch.randn(B * N, C_, generator=g) * 0.8
    if adversarial and N >= 256:
        k[N - 100, :D] = q[7, :D] * 4                   # late spikes (frame 0, head 0)
        k[N - 37, :D] = q[40, :D] * 4
        k[:64, :D] = -q[9, :D] * 3                      # query 9: first key tile far below the later ones
        q[11, :D] = 0.0                                 # flat row
        k[:, D:2 * D] -= 2.5 * torch.sign(q[13, D:2 * D])     # head 1, query 13: every score strongly negative
        k[N - 70, :D] = q[50, :D] * 60                  # a spike far past exp2's range relative to the anchor (inf in the speculative pass)
        k[130:190, :D] += q[60, :D] * 0.9               # a whole tile moderately above the running max of query 60: the row-sum trigger without a single large p
        if H > 2:
            k[:64, 2 * D:3 * D] = -q[70, 2 * D:3 * D] * 40   # head 2, query 70: anchor tile hundreds below the rest
    qkv = torch.cat([(q * f), k, v], 1).half().to(DEV)
    hip.TIMED = []
    o = run(qkv, B, H, N)
    name = hip.TIMED[0][1]["kernel"]
    hip.TIMED = None
    kq = lambda t: t.float().reshape(B, N, H, D).permute(0, 2, 1, 3)
    qf, kf, vf = kq(qkv[:, :C_]), kq(qkv[:, C_:2 * C_]), kq(qkv[:, 2 * C_:])
    got = o.float().reshape(B, N, H, D).permute(0, 2, 1, 3)
    num = den = 0.0
    worst = 0.0
    for b in range(B):
        ref = torch.softmax((qf[b].double() @ kf[b].double().transpose(-1, -2)) * 0.6931471805599453, dim=-1) @ vf[b].double()
        d = got[b].double() - ref
        num += float((d ** 2).sum())
        den += float((ref ** 2).sum())
        worst = max(worst, float(d.abs().max()))
    rel = (num / den) ** 0.5
    ok = bool(torch.isfinite(o).all()) and rel < 1e-3 and worst < 2e-2
    print(f"check B={B} H={H:2d} N={N:5d} adv={int(adversarial)} kernel={name}: rel {rel:.3e} worst {worst:.3e} {'OK' if ok else 'FAIL'}", flush=True)
    return ok


def bench():
    e0, e1 = hip.Event(), hip.Event()
    tot = 0.0
    for name, B, H, N in [("self 64^2 x16", 16, 5, 4096), ("self 32^2 x16", 16, 10, 1024), ("self 16^2 x16", 16, 20, 256),
                          ("self 64^2 x8", 8, 5, 4096), ("self 32^2 x8", 8, 10, 1024)]:
        C_ = H * 64
        qkv = torch.randn(B * N, 3 * C_, device=DEV)
        qkv[:, :C_] *= 64 ** -0.5 * LOG2E
        qkv = qkv.half()
        for _ in range(3):
            run(qkv, B, H, N)
        e0.record()
        for _ in range(20):
            run(qkv, B, H, N)
        e1.record()
        e1.sync()
        us = 1e3 * e0.elapsed_ms(e1) / 20
        tf = 4.0 * B * H * N * N * 64 / (us * 1e-6) / 1e12
        print(f"bench {name:14s} {us:9.2f} us  {tf:7.1f} TF/s  ({tf / 2500:.3f} of 2.5 PF)", flush=True)


if __name__ == "__main__":
    hip.lib()
    print("variant: " + " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("MGLD_ATTN")), flush=True)
    ok = True
    for shp in [(1, 2, 64), (3, 5, 256), (2, 3, 192), (8, 10, 1024), (8, 5, 4096), (1, 1, 128), (2, 2, 320)]:
        ok = check(*shp) and ok
    ok = check(4, 5, 4096, adversarial=False, seed=5) and ok
    print("ALL OK" if ok else "SOME FAILED", flush=True)
    bench()
