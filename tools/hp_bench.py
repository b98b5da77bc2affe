#!/usr/bin/env python
"""GPU microbench of the first-stage encode (mgld_vsr_amd/vae.py::AutoencoderKL.encode): MGLD_HP_ENCODER=1 (high precision, default) vs 0
(the fp16 encoder), for `frames` 512^2 frames.   python tools/hp_bench.py [frames ...]
Under rocprofv3 --kernel-trace --stats the per-kernel split of the same call comes out (tools/lease.sh hpprof)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    frames = [int(a) for a in sys.argv[1:]] or [8]
    from mgld_vsr_amd import vae
    from mgld_vsr_amd.pipeline import VSRPipeline, model_configs
    pipe = VSRPipeline(num_frames=8, ddpm_steps=4, configs=model_configs(8))
    m = pipe.model
    for T in frames:
        x = (torch.rand(T, 3, 512, 512, generator=torch.Generator().manual_seed(1)) * 2 - 1).cuda()
        for _ in range(2):
            m.encode_first_stage(x)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        a.record()
        for _ in range(n):
            m.encode_first_stage(x)
        b.record()
        torch.cuda.synchronize()
        print(f"HP_ENCODER={int(vae.HP_ENCODER)} frames={T}: {a.elapsed_time(b) / n:.2f} ms per encode, arena {pipe.engine().arena.bytes_reserved() / 2**30:.1f} GiB", flush=True)


if __name__ == "__main__":
    main()
