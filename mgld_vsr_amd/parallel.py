"""Multi-GPU host logic: one process per GPU, segments sharded over ranks.

The reference scales inference by running N processes that each take `seq_idx % n_gpus == select_idx`
(scripts/vsr_val_ddpm_text_T_vqganfin_oldcanvas_tile.py:292-293,337-339); segments of `n_frames` frames are
independent (no state crosses them), so the data path needs NO collective.  torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests) is used only for the timing barrier, the
max-over-ranks reduction of the measured time and an optional gather of per-rank results.
"""
import os

import torch


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend="nccl", device=None):
    """Initialise the default process group from the torchrun environment (RANK / WORLD_SIZE / MASTER_*)."""
    import torch.distributed as dist
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_segments(n_segments, rank, world):
    """Indices of the segments rank `rank` owns: the reference's round-robin rule, applied to segments."""
    return [i for i in range(n_segments) if i % world == rank]


def segment_bounds(n_frames_total, n_frames):
    """(start, stop, n_pad) per segment: frame lists are padded to a multiple of n_frames by repeating the last frame
    (oldcanvas_tile.py:345-346)."""
    n_pad = (-n_frames_total) % n_frames
    total = n_frames_total + n_pad
    return [(s, s + n_frames, max(0, s + n_frames - n_frames_total)) for s in range(0, total, n_frames)]


def barrier(sync_device=True):
    import torch.distributed as dist
    if sync_device and torch.cuda.is_available():
        torch.cuda.synchronize()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
        if sync_device and torch.cuda.is_available():
            torch.cuda.synchronize()


def max_over_ranks(value, device=None):
    """max-reduce a python float over the ranks (the bench reports the slowest rank's time)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def gather_frames(local_frames, segment_ids, n_segments, device=None):
    """Collect per-rank outputs on rank 0: `local_frames` = list of [T,3,H,W] tensors for `segment_ids`.
    Returns the ordered list on rank 0, None elsewhere.  (Not on the timed path: the reference lets every process write
    its own PNGs.)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        out = [None] * n_segments
        for i, f in zip(segment_ids, local_frames):
            out[i] = f
        return out
    payload = [(int(i), f.cpu()) for i, f in zip(segment_ids, local_frames)]
    gathered = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(payload, gathered, dst=0)
    if dist.get_rank() != 0:
        return None
    out = [None] * n_segments
    for part in gathered:
        for i, f in part:
            out[i] = f
    return out
