#!/usr/bin/env python
"""Phase of the segments in flight: times K segments through pipeline.SegmentPool for a list of entry offsets between the workers
(SegmentPool.stagger_ms), one process, one model build, alternating over the list `--reps` times.  Scratch tool — not product, not a test.

    python tools/stagger_sweep.py --staggers 0,2,4,8,13,20,26 --segments 12 --reps 2 [--inflight 3]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--staggers", default="0,4,8,13,20")
    ap.add_argument("--segments", type=int, default=12)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--inflight", type=int, default=3)
    ap.add_argument("--frames", type=int, default=8)
    a = ap.parse_args()
    sys.argv = [sys.argv[0], "--frames", str(a.frames)]
    import bench
    from mgld_vsr_amd.pipeline import SegmentPool
    args = bench.parse()
    torch.cuda.set_device(0)
    pipe = bench.build_pipeline(args)
    k = a.inflight
    pool = SegmentPool(lambda: bench.build_pipeline(args), k, first=pipe)
    ins = [bench.make_inputs(pool.pipes[i], args, i) for i in range(k)]
    jobs = [((ins[j % k][0],), dict(flows=ins[j % k][2], masks=ins[j % k][3], noise=ins[j % k][1], tile=None, use_graph=True))
            for j in range(a.segments)]
    for i in range(k):
        pool.run_on(i, [jobs[i]] * 2)
    vals = [float(v) for v in a.staggers.split(",")]
    res = {v: [] for v in vals}
    for _ in range(a.reps):
        for v in vals:
            pool.stagger_ms = v
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pool.run(jobs)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res[v].append(1e3 * dt / a.segments)
            lat = sorted(pool.last_latency_ms.values())
            print(f"stagger {v:5.1f} ms: {1e3 * dt / a.segments:7.2f} ms/segment  {a.frames * a.segments / dt:6.3f} frames/s  latency median {lat[len(lat) // 2]:.0f} ms", flush=True)
    print("best of reps: " + "  ".join(f"{v:g}: {min(r):.1f}" for v, r in res.items()))


if __name__ == "__main__":
    main()
