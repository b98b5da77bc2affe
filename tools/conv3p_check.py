#!/usr/bin/env python
"""Scratch check of the patch-staged 3x3 conv (conv3p) against torch conv2d; prints the launcher's config code per case."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mgld_vsr_amd import hip  # noqa: E402
from mgld_vsr_amd.engine import pack_conv3x3, tile_conv3p  # noqa: E402

DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def tok(x):
    n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(n * h * w, c).contiguous()


def rel(a, b):
    return float((a - b).norm() / b.norm())


def main():
    hip.lib()
    hip.ensure_workspace()
    bad = 0
    cases = [(2, 128, 96, 16, 16, False, 0), (4, 320, 640, 64, 64, True, 0), (8, 640, 640, 32, 32, True, 1), (8, 1280, 1280, 16, 16, True, 2),
             (2, 64, 320, 32, 32, False, 1), (3, 96, 64, 16, 16, False, 2), (1, 320, 320, 24, 16, True, 0), (8, 320, 320, 64, 64, True, 1),
             (2, 2560, 1280, 16, 16, True, 0), (1, 32, 40, 16, 16, False, 0)]
    for n, cin, cout, h, w, ti, epi in cases:
        x = rnd(n, cin, h, w, seed=1).half()
        wt = (rnd(cout, cin, 3, 3, seed=2) * (9 * cin) ** -0.5).half()
        b = rnd(cout, seed=3)
        ref = F.conv2d(x.float(), wt.float(), b, padding=1)
        wk = (pack_conv3x3(wt, tap_inner=True) if ti else wt.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous()).to(DEV)
        xt = tok(x).to(DEV)
        kw = {}
        if epi == 1:     # residual + SiLU + alpha/beta
            r = rnd(n * h * w, cout, seed=4).half()
            kw = dict(resid=r.to(DEV), act=hip.ACT_SILU, alpha=0.5, beta=2.0)
            ref = 0.5 * F.silu(ref) + 2.0 * r.float().reshape(n, h, w, cout).permute(0, 3, 1, 2)
        elif epi == 2:   # per-frame row vector, strided input (concat slice)
            emb = rnd(n, cout, seed=5)
            kw = dict(rowvec=emb.to(DEV), rows_per_frame=h * w)
            ref = ref + emb[:, :, None, None]
            big = torch.zeros(n * h * w, cin + 64, dtype=torch.half, device=DEV)
            big[:, 32:32 + cin] = xt
            xt = big[:, 32:32 + cin]
        out = torch.empty(n * h * w, cout, dtype=torch.half, device=DEV)
        hip.IGEMM_LOG = []
        hip.igemm(xt, wk, out, mode=hip.MODE_CONV3X3, bias=b.to(DEV), conv=(cin, h, w, h, w, 1, 1, 1, 0), tap_inner=1 if ti else 0, **kw)
        torch.cuda.synchronize()
        assert hip.conv3p_applies(n, cin, cout, h, w)
        out2 = torch.empty_like(out)
        hip.igemm(xt, tile_conv3p(wk, cin, ti), out2, mode=hip.MODE_CONV3X3, bias=b.to(DEV), conv=(cin, h, w, h, w, 1, 1, 1, 0),
                  tap_inner=2, N=cout, K=9 * cin, **kw)
        torch.cuda.synchronize()
        same = torch.equal(out, out2)
        cfg = hip.igemm_config(hip.IGEMM_LOG[0])
        hip.IGEMM_LOG = None
        o = out.cpu().float().reshape(n, h, w, cout).permute(0, 3, 1, 2)
        e = rel(o, ref)
        ok = e < 1e-3 and same
        bad += not ok
        print(f"n={n} cin={cin} cout={cout} {h}x{w} tap_inner={ti} epi={epi} cfg={cfg} rel_l2={e:.2e} tiled_identical={same} {'ok' if ok else 'FAIL'}", flush=True)
    print("FAILED" if bad else "ALL OK")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
