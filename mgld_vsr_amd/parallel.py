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


def gather_floats(value, device=None):
    """all-gather one python float per rank -> list indexed by rank (bench.py reports every rank's ms per step, not only the max)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [round(float(value), 2)]
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [round(float(o[0]), 2) for o in out]


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


# ----------------------------------------------------------------------------------------------------------------------
# Intra-segment frame sharding (second scheme): the T frames of ONE segment are split over the ranks.
# Frames are independent except at three coupling points, each served by one exchange over RCCL/xGMI:
#   * SpatialTemporalConv (UNet mid block x2, VAE video decoder x13): +-1-frame halo  -> neighbour send/recv
#   * TemporalAttention (UNet mid block): every frame attends to all T frames         -> all-gather of q|k|v rows
#   * motion guidance (compute_temporal_condition_v4): chain over neighbouring frames -> all-gather of the latents
# Zero padding at the two ends of the clip (Conv3d padding, diffusionmodules/util.py:298) is kept on the first / last rank.
# ----------------------------------------------------------------------------------------------------------------------
class DistComm:
    """torch.distributed transport ("nccl" = RCCL on the GPU box, "gloo" in the CPU tests).

    DistComm.measure = True (bench.py's sharded modes): every exchange is bracketed by a device event pair on the current stream and its
    payload counted; DistComm.report() returns what this rank sent / received and how long the exchanges took, so the bench line can put
    comm_plan()'s predicted bytes next to measured bytes and time."""
    measure = False
    _log = []           # (bytes this rank contributes or receives, event pair or host seconds)

    @classmethod
    def _bracket(cls, nbytes, fn, device):
        if not cls.measure:
            return fn()
        if device.type == "cuda":
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn()
            e1.record()
            cls._log.append((nbytes, (e0, e1)))
        else:
            import time
            t0 = time.perf_counter()
            r = fn()
            cls._log.append((nbytes, time.perf_counter() - t0))
        return r

    @classmethod
    def report(cls, reset=True):
        """{"calls", "bytes", "ms"} over the exchanges logged since the last reset (synchronises the device)"""
        if any(not isinstance(t, float) for _, t in cls._log):
            torch.cuda.synchronize()
        ms = sum((1e3 * t) if isinstance(t, float) else t[0].elapsed_time(t[1]) for _, t in cls._log)
        out = {"calls": len(cls._log), "bytes": int(sum(b for b, _ in cls._log)), "ms": round(ms, 3)}
        if reset:
            cls._log = []
        return out

    def all_gather(self, t, shard, out=None):
        """`out`: a caller-owned buffer of the gathered shape (fixed address: what the graph pieces of a sharded step need)"""
        import torch.distributed as dist
        t = t.contiguous()
        if out is None:
            out = torch.empty((shard.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        assert out.is_contiguous() and out.shape[0] == shard.world * t.shape[0]

        def run():
            try:
                dist.all_gather_into_tensor(out, t)
            except (RuntimeError, NotImplementedError):
                dist.all_gather(list(out.chunk(shard.world, 0)), t)
        self._bracket(t.numel() * t.element_size() * max(1, shard.world - 1), run, t.device)     # received from the other ranks
        return out

    def exchange(self, x, rows_per_frame, recv_left, recv_right, shard):
        """x: [F*rows_per_frame, C] local frames.  recv_left <- last frame of rank-1, recv_right <- first frame of rank+1
        (zeros at the ends of the clip)."""
        import torch.distributed as dist
        ops, keep = [], []
        if shard.rank > 0:
            first = x[:rows_per_frame].contiguous()
            keep.append(first)
            ops += [dist.P2POp(dist.isend, first, shard.rank - 1), dist.P2POp(dist.irecv, recv_left, shard.rank - 1)]
        else:
            recv_left.zero_()
        if shard.rank < shard.world - 1:
            last = x[x.shape[0] - rows_per_frame:].contiguous()
            keep.append(last)
            ops += [dist.P2POp(dist.isend, last, shard.rank + 1), dist.P2POp(dist.irecv, recv_right, shard.rank + 1)]
        else:
            recv_right.zero_()
        if ops:
            def run():
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            self._bracket(sum(k.numel() * k.element_size() for k in keep), run, x.device)               # sent to the neighbours


    def gather_tiles(self, t, shard, out=None):
        """[per*rows, ...] (this rank's tiles, zero-padded to `per` tiles) -> [world*per*rows, ...] in rank order"""
        return self.all_gather(t, shard, out=out)


class RecordingComm:
    """World-size-1 transport that keeps a copy of every tensor handed to it: the full-clip reference trace that
    ReplayComm serves to a virtual rank (single-GPU validation of the sharded math, tests/test_nets_gpu.py)."""

    def __init__(self):
        self.trace = []

    def all_gather(self, t, shard, out=None):
        self.trace.append(("gather", t.detach().clone()))
        if out is None:
            return t
        out.copy_(t)
        return out

    def exchange(self, x, rows_per_frame, recv_left, recv_right, shard):
        self.trace.append(("halo", x.detach().clone()))
        recv_left.zero_()
        recv_right.zero_()

    def gather_tiles(self, t, shard, out=None):
        self.trace.append(("tiles", t.detach().clone()))
        if out is None:
            return t
        out.copy_(t)
        return out


class ReplayComm:
    """Serves virtual rank `shard.rank` of `shard.world` from a RecordingComm trace of the full clip and checks, call by
    call, that what this rank contributes equals its slice of the full-clip tensors (max relative deviation kept in
    `worst`)."""

    def __init__(self, trace):
        self.trace, self.pos, self.worst = trace, 0, 0.0

    def _next(self, kind, local, shard, unit):
        k, full = self.trace[self.pos]
        self.pos += 1
        assert k == kind, f"communication sequence diverged: expected {k}, got {kind}"
        mine = full[shard.rank * local.shape[0]:(shard.rank + 1) * local.shape[0]]
        assert mine.shape == local.shape, (mine.shape, local.shape)
        den = float(mine.float().norm()) + 1e-30
        self.worst = max(self.worst, float((local.float() - mine.float()).norm()) / den)
        return full

    def all_gather(self, t, shard, out=None):
        full = self._next("gather", t, shard, None)
        if out is None:
            return full.clone()
        out.copy_(full)
        return out

    def exchange(self, x, rows_per_frame, recv_left, recv_right, shard):
        full = self._next("halo", x, shard, rows_per_frame)
        lo = shard.rank * x.shape[0]
        hi = lo + x.shape[0]
        if lo > 0:
            recv_left.copy_(full[lo - rows_per_frame:lo])
        else:
            recv_left.zero_()
        if hi < full.shape[0]:
            recv_right.copy_(full[hi:hi + rows_per_frame])
        else:
            recv_right.zero_()


    def gather_tiles(self, t, shard, out=None):
        k, full = self.trace[self.pos]
        self.pos += 1
        assert k == "tiles", f"communication sequence diverged: expected {k}, got tiles"
        rows = t.shape[0] // shard.per                       # rows of one tile
        valid = (shard.k1 - shard.k0) * rows
        mine = full[shard.k0 * rows:shard.k0 * rows + valid]
        den = float(mine.float().norm()) + 1e-30
        self.worst = max(self.worst, float((t[:valid].float() - mine.float()).norm()) / den)
        if out is None:
            out = torch.zeros((shard.world * shard.per * rows,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        else:
            out.zero_()
        out[:full.shape[0]] = full
        return out


class TileShard:
    """Aggregation sampling (sample_canvas, ddpm.py:4191-4322) over ranks — BASELINE configs[3], SURVEY 8(e) "Tiled path": the
    overlapping latent tiles of a step are independent (struct-cond encoder + UNet) clips.  Rank r evaluates tiles [k0, k1) of the
    reference's tile order; ONE all-gather per step brings every tile's eps to every rank, which then stitches the canvas with the
    Gaussian weights and runs the (tiny) posterior + guidance on the whole canvas, replicated.  Tiles are dealt in contiguous
    blocks of `per` = ceil(n_tiles / world); the last ranks' missing tiles are zero padding in the exchange."""

    def __init__(self, n_tiles, rank, world, comm=None):
        self.n_tiles, self.rank, self.world = int(n_tiles), int(rank), int(world)
        self.per = -(-self.n_tiles // self.world)
        self.k0 = min(self.rank * self.per, self.n_tiles)
        self.k1 = min(self.k0 + self.per, self.n_tiles)
        self.comm = comm if comm is not None else DistComm()

    def slot(self, k):
        """position of global tile k in the gathered buffer (in tiles)"""
        return (k // self.per) * self.per + (k % self.per)

    def gather(self, t, out=None):
        return self.comm.gather_tiles(t, self, out=out)


class FrameShard:
    """Frames [f0, f1) of a T-frame segment live on this rank (T % world == 0)."""

    def __init__(self, T, rank, world, comm=None):
        if T % world:
            raise ValueError(f"frame sharding needs the segment length ({T}) to be a multiple of the rank count ({world})")
        self.T, self.rank, self.world = int(T), int(rank), int(world)
        self.F = self.T // self.world
        self.f0, self.f1 = self.rank * self.F, (self.rank + 1) * self.F
        self.comm = comm if comm is not None else DistComm()

    def local(self, t, dim=0):
        """this rank's frames of a full-clip tensor"""
        return t.narrow(dim, self.f0, self.F)

    def all_gather(self, t, out=None):
        """[F*k, ...] per rank -> [T*k, ...] in rank (= frame) order (into `out` when given)"""
        return self.comm.all_gather(t, self, out=out)

    def halo(self, x, rows_per_frame, recv_left, recv_right):
        self.comm.exchange(x, rows_per_frame, recv_left, recv_right, self)


# ----------------------------------------------------------------------------------------------------------------------
# Communication plan of the sharded modes: what crosses xGMI, how often, how many bytes (DESIGN.md section 4 quotes these
# figures; bench.py --spawn-selftest replays the plan with dummy tensors over the real process group: launcher + transport
# check without a kernel, gloo on CPU / RCCL on GPUs).
# ----------------------------------------------------------------------------------------------------------------------
def comm_plan(mode, T=8, H=512, W=512, world=8, steps=50, n_tiles=9, ch=128, ch_mult=(1, 2, 4, 4), unet_mid_ch=1280):
    """-> {"per_step": [...], "per_segment": [...], totals}: every exchange of ONE rank in mode
    'segment' (no data-path exchange), 'frame' (FrameShard: halo send/recv + all-gathers) or 'tile' (TileShard: one all-gather per
    step + the frame-split VAE).  Entry: (what, kind, count, bytes sent by this rank per occurrence).  Shipped geometry: latent
    H/8 x W/8 x 4 (fp32), UNet mid block at 1/64 resolution with 1280 channels (fp16), video decoder levels ch * ch_mult."""
    F = max(1, T // world)
    h, w = H // 8, W // 8
    per_step, per_seg = [], []
    if mode == "segment" or world == 1:
        return {"mode": mode, "per_step": [], "per_segment": [], "bytes_per_step": 0, "bytes_per_segment": 0, "steps": steps}
    if mode == "frame":
        mid = (h // 8) * (w // 8) * unet_mid_ch * 2                       # one frame of the UNet's 1/64-resolution level, fp16
        per_step.append(("unet mid-block SpatialTemporalConv halo (x2, both neighbours)", "p2p", 2, 2 * mid))
        per_step.append(("TemporalAttention q|k|v all-gather", "all_gather", 1, F * 3 * mid))
        per_step.append(("guidance: latent all-gather", "all_gather", 1, F * 4 * h * w * 4))
    elif mode == "tile":
        per = -(-n_tiles // world)
        per_step.append(("eps tiles all-gather", "all_gather", 1, per * T * 4 * 64 * 64 * 4))
    # the frame-split video VAE decode (both sharded modes): one-frame halos of the 13 temporal convolutions, fp16
    lv = [ch * m for m in ch_mult]
    halos = [("decoder mid temporal_mixing", h * w * lv[-1] * 2)]
    res = (h, w)
    for i in reversed(range(len(ch_mult))):
        halos += [(f"decoder up.{i} temporal_mixing x3", 3 * res[0] * res[1] * lv[i] * 2)]
        if i:
            res = (2 * res[0], 2 * res[1])
    if mode == "frame" or T % world == 0:
        for name, b in halos:
            per_seg.append((name + " halo (both neighbours)", "p2p", 1, 2 * b))
        per_seg.append(("init-latent all-gather (first-stage encode split)", "all_gather", 1, F * 4 * h * w * 4))
        per_seg.append(("HR frames all-gather", "all_gather", 1, F * 3 * H * W * 4))
    bs = sum(c * b for _, _, c, b in per_step)
    bg = sum(c * b for _, _, c, b in per_seg)
    return {"mode": mode, "T": T, "world": world, "per_step": per_step, "per_segment": per_seg, "bytes_per_step": bs,
            "bytes_per_segment": bs * steps + bg, "steps": steps}


def comm_dry_run(plan, device="cpu", step_repeats=2):
    """replay the exchanges of `plan` (comm_plan) with dummy tensors over the initialised process group, through the same DistComm
    calls the sharded modes make; returns the bytes this rank sent.  No kernels: a launcher / rendezvous / transport check."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    rank, world = dist.get_rank(), dist.get_world_size()
    comm = DistComm()
    sh = FrameShard(world, rank, world, comm) if plan["mode"] != "tile" else TileShard(world, rank, world, comm)
    sent = 0

    def run(entries, reps):
        nonlocal sent
        for _ in range(reps):
            for what, kind, count, nbytes in entries:
                n = max(1, min(nbytes, 1 << 20) // 2)          # transport check: at most 1 MiB per message
                for _ in range(count):
                    if kind == "all_gather":
                        t = torch.full((1, n), float(rank), dtype=torch.float16, device=device)
                        out = comm.all_gather(t, sh)
                        assert out.shape[0] == world and float(out[world - 1, 0]) == world - 1
                    else:
                        x = torch.full((2, n), float(rank), dtype=torch.float16, device=device)
                        left, right = torch.empty(1, n, dtype=torch.float16, device=device), torch.empty(1, n, dtype=torch.float16, device=device)
                        comm.exchange(x, 1, left, right, sh)
                        assert float(left[0, 0]) == (rank - 1 if rank > 0 else 0) and float(right[0, 0]) == (rank + 1 if rank < world - 1 else 0)
                    sent += nbytes
    run(plan["per_step"], step_repeats)
    run(plan["per_segment"], 1)
    return sent
