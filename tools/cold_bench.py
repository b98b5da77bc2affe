#!/usr/bin/env python
"""Does the ADDRESS HISTORY of a buffer matter?  GroupNorm apply (low-plane form) and a copy over (a) one hot buffer set, (b) a ring of cold
sets, (c) cold inputs / one recycled output, (d) chain: input = the previous launch's output (hot), output cold (what a bump allocator gives a
step), (e) chain through a ring of 3 recycled buffers.  Scratch tool for the arena-recycling question — not product, not a test."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mgld_vsr_amd import hip  # noqa: E402


def run(fn_list, e0, e1, reps=3):
    best = 1e30
    for _ in range(reps):
        e0.record()
        for fn in fn_list:
            fn()
        e1.record()
        e1.sync()
        best = min(best, 1e3 * e0.elapsed_ms(e1) / len(fn_list))
    return best


def main():
    hip.lib()
    dev = "cuda"
    e0, e1 = hip.Event(), hip.Event()
    R = int(os.environ.get("RING", "24"))
    for frames, rows, C in [(16, 4096, 320), (16, 4096, 640), (16, 1024, 1280)]:
        n = frames * rows
        mb = n * C * 2 / 1e6
        xs = [torch.randn(n, C, device=dev).half() for _ in range(R)]
        ls = [torch.randn(n, C, device=dev).half() for _ in range(R)]
        ys = [torch.empty(n, C, device=dev, dtype=torch.float16) for _ in range(R)]
        gs = torch.empty(frames, hip.gn_chunks(rows), 32, 2, dtype=torch.float64, device=dev)
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        hip.gn_stats(xs[0], frames, rows, 32, gs)

        def ap(x, l, y):
            return lambda: hip.gn_apply(x, gs, 1e-5, g, b, y, frames, rows, 32, True, x_lo=l)

        def ap1(x, y):
            return lambda: hip.gn_apply(x, gs, 1e-5, g, b, y, frames, rows, 32, True)

        K = 48
        res = {}
        res["lo hot"] = run([ap(xs[0], ls[0], ys[0])] * K, e0, e1)
        res["lo cold-all"] = run([ap(xs[i % R], ls[i % R], ys[i % R]) for i in range(K)], e0, e1)
        res["lo cold-in hot-out"] = run([ap(xs[i % R], ls[i % R], ys[0]) for i in range(K)], e0, e1)
        res["lo hot-in cold-out"] = run([ap(xs[0], ls[0], ys[i % R]) for i in range(K)], e0, e1)
        # one-plane chains: y[i] = f(y[i-1])
        res["chain cold-out (bump)"] = run([ap1(ys[i % R], ys[(i + 1) % R]) for i in range(K)], e0, e1)
        res["chain ring3 (recycled)"] = run([ap1(ys[i % 3], ys[(i + 1) % 3]) for i in range(K)], e0, e1)
        res["1p hot"] = run([ap1(xs[0], ys[0])] * K, e0, e1)
        print(f"frames={frames} rows={rows} C={C} ({mb:.1f} MB per plane, ring {R} = {R * mb * 3 / 1e3:.1f} GB)")
        for k, v in res.items():
            planes = 3 if k.startswith("lo") else 2
            print(f"   {k:26s} {v:8.2f} us  {planes * mb / v:6.2f} TB/s")
        del xs, ls, ys
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
