#!/usr/bin/env python
"""One-rank RCCL self-test of the torch.distributed calls the multi-GPU paths make (init with device_id, barrier, max-reduce,
all_gather_into_tensor through parallel.DistComm).  Run under torchrun on a GPU box:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/nccl_selftest.py
Scratch tool (the 2-rank semantics are covered by the gloo tests; 8-GPU runs belong to the driver)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgld_vsr_amd import parallel  # noqa: E402

rank, world, local = parallel.env_rank_world()
torch.cuda.set_device(local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
parallel.barrier()
print("max_over_ranks", parallel.max_over_ranks(1.5 + rank))
sh = parallel.FrameShard(4 * world, rank, world)
x = torch.arange(4 * 3, dtype=torch.float16, device="cuda").reshape(4, 3) + 100 * rank
g = sh.all_gather(x)
assert g.shape == (4 * world, 3) and torch.equal(g[rank * 4:(rank + 1) * 4], x)
l, r = torch.ones(2, 3, device="cuda", dtype=torch.float16), torch.ones(2, 3, device="cuda", dtype=torch.float16)
sh.halo(x, 2, l, r)
if world == 1:
    assert float(l.abs().sum()) == 0 and float(r.abs().sum()) == 0
parallel.barrier()
dist.destroy_process_group()
print("rccl self-test ok: world", world)
