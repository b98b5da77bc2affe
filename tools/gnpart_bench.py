#!/usr/bin/env python
"""Microbench: what the statistics output of the ping-pong patch convolution (MgldIGemm.gn_part) costs per launch, against the
mgld_gn_stats launch it replaces.  Scratch tool — not product, not a test.
    python tools/gnpart_bench.py [--iters 20]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mgld_vsr_amd import hip  # noqa: E402
from mgld_vsr_amd.engine import tile_conv3p  # noqa: E402

# (name, frames, H, W, Cin, Cout)
SHAPES = [("unet64 320->320", 8, 64, 64, 320, 320), ("unet64 640->320", 8, 64, 64, 640, 320), ("unet32 640->640", 8, 32, 32, 640, 640),
          ("unet32 1280->640", 8, 32, 32, 1280, 640), ("vae128 512->512", 8, 128, 128, 512, 512), ("vae256 256->256", 8, 256, 256, 256, 256),
          ("vae512 128->128", 8, 512, 512, 128, 128)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    hip.lib()
    hip.ensure_workspace()
    e0, e1 = hip.Event(), hip.Event()

    def timeit(fn):
        best = 1e30
        for _ in range(3):
            fn()
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            e1.sync()
            best = min(best, 1e3 * e0.elapsed_ms(e1) / args.iters)
        return best
    for name, n, h, w, cin, cout in SHAPES:
        x = torch.randn(n * h * w, cin, device="cuda").half()
        wt = tile_conv3p((torch.randn(cout, 9 * cin, device="cuda") * (9 * cin) ** -0.5).half(), cin, False)
        b = torch.randn(cout, device="cuda")
        out = torch.empty(n * h * w, cout, dtype=torch.half, device="cuda")
        kw = dict(mode=hip.MODE_CONV3X3, conv=(cin, h, w, h, w, 1, 1, 1, 0), tap_inner=2, N=cout, K=9 * cin, bias=b)
        holder = []

        def part(chunks):
            if not holder:
                holder.append(torch.empty(n * chunks, 2, cout, dtype=torch.float32, device="cuda"))
            return holder[0]
        t0 = timeit(lambda: hip.igemm(x, wt, out, **kw))
        t1 = timeit(lambda: hip.igemm(x, wt, out, gn_part=part, **kw))
        gs = torch.empty(n, hip.gn_chunks(h * w), 32, 2, dtype=torch.float64, device="cuda")
        t2 = timeit(lambda: hip.gn_stats(out, n, h * w, 32, gs))
        print(f"{name:18s} conv {t0:8.1f} us   conv + statistics {t1:8.1f} us (+{t1 - t0:6.1f})   mgld_gn_stats {t2:7.1f} us   chunks {holder[0].shape[0] // n if holder else 0}", flush=True)


if __name__ == "__main__":
    main()
