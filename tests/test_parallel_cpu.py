"""Multi-process (world_size 2, gloo, CPU) test of the N>1 host path: segment sharding, barrier, max-over-ranks timing
reduction and result gathering — the same functions bench.py uses with RCCL on the GPU box."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from mgld_vsr_amd import parallel
    r, w, _ = parallel.init(backend="gloo")
    assert (r, w) == (rank, world)
    n_seg = 7
    mine = parallel.shard_segments(n_seg, r, w)
    # every rank "processes" its segments: the result encodes (segment id, rank)
    frames = [torch.full((2, 3, 4, 4), float(10 * i + r)) for i in mine]
    parallel.barrier(sync_device=False)
    dt = parallel.max_over_ranks(1.0 + rank, device="cpu")
    out = parallel.gather_frames(frames, mine, n_seg)
    if rank == 0:
        ret["dt"] = dt
        ret["owners"] = [int(f[0, 0, 0, 0]) % 10 for f in out]
        ret["ids"] = [int(f[0, 0, 0, 0]) // 10 for f in out]
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_segment_sharding():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret["dt"] == 2.0                                   # max over ranks
    assert ret["ids"] == list(range(7))                       # every segment processed exactly once, in order
    assert ret["owners"] == [i % world for i in range(7)]     # the reference's round-robin rule


def test_segment_bounds_and_shards():
    sys.path.insert(0, ROOT)
    from mgld_vsr_amd import parallel
    assert parallel.segment_bounds(12, 5) == [(0, 5, 0), (5, 10, 0), (10, 15, 3)]
    assert parallel.segment_bounds(10, 5) == [(0, 5, 0), (5, 10, 0)]
    got = sorted(sum((parallel.shard_segments(13, r, 4) for r in range(4)), []))
    assert got == list(range(13))
    assert parallel.shard_segments(3, 5, 8) == []


# ----------------------------------------------------------------------------------------------------------------------
# intra-segment frame sharding (parallel.FrameShard): halo exchange + all-gather over gloo, checked against the full clip
# ----------------------------------------------------------------------------------------------------------------------
def _tconv_full(x, w):
    """3-tap conv over the frame axis with zero padding: x [T, hw, C], w [3, C, C]"""
    T = x.shape[0]
    z = torch.zeros_like(x[:1])
    ext = torch.cat([z, x, z], 0)
    return sum(ext[dt:dt + T] @ w[dt] for dt in range(3))


def _shard_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from mgld_vsr_amd import parallel
    parallel.init(backend="gloo")
    T, hw, C = 6, 5, 4
    g = torch.Generator().manual_seed(7)
    x = torch.randn(T, hw, C, generator=g)
    w = torch.randn(3, C, C, generator=g)
    sh = parallel.FrameShard(T, rank, world)
    parallel.DistComm.measure = True          # exchange accounting (bytes + time per call: what bench.py prints beside comm_plan)
    parallel.DistComm.report()
    xl = sh.local(x).reshape(sh.F * hw, C).contiguous()
    # temporal conv on a halo-extended copy, exactly as Engine.tconv3 lays it out
    ext = torch.empty((sh.F + 2) * hw, C)
    ext[hw:(sh.F + 1) * hw] = xl
    sh.halo(xl, hw, ext[:hw], ext[(sh.F + 1) * hw:])
    e3 = ext.reshape(sh.F + 2, hw, C)
    yl = sum(e3[dt:dt + sh.F] @ w[dt] for dt in range(3))
    y = sh.all_gather(yl.reshape(sh.F * hw, C)).reshape(T, hw, C)
    ok_conv = torch.allclose(y, _tconv_full(x, w), atol=1e-5)
    # all-gather keeps rank (= frame) order
    buf = torch.full((T, 1), -1.0)     # caller-owned destination (what the graph pieces of a sharded step use): filled in place
    ids = sh.all_gather(torch.arange(sh.f0, sh.f1, dtype=torch.float32).reshape(sh.F, 1), out=buf)
    ok_order = ids is buf and ids.flatten().tolist() == list(range(T))
    rep = parallel.DistComm.report()
    parallel.DistComm.measure = False
    ret[rank] = (bool(ok_conv), bool(ok_order), float(ext[:hw].abs().sum()), float(ext[(sh.F + 1) * hw:].abs().sum()), rep)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_frame_shard_halo_and_gather(world):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_shard_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        ok_conv, ok_order, left, right, rep = ret[r]
        assert ok_conv and ok_order
        assert rep["calls"] == 3 and rep["bytes"] > 0 and rep["ms"] >= 0.0, rep      # one halo exchange + two all-gathers were logged
        assert (left == 0.0) == (r == 0)              # zero padding only at the two ends of the clip
        assert (right == 0.0) == (r == world - 1)


def test_frame_shard_replay_matches_recording():
    """RecordingComm / ReplayComm (the single-GPU validation transport) reproduce what real ranks would exchange."""
    sys.path.insert(0, ROOT)
    from mgld_vsr_amd import parallel
    T, hw, C = 4, 3, 2
    x = torch.randn(T * hw, C)
    rec = parallel.RecordingComm()
    s1 = parallel.FrameShard(T, 0, 1, rec)
    l, r = torch.ones(hw, C), torch.ones(hw, C)
    s1.halo(x, hw, l, r)
    assert float(l.abs().sum()) == 0 and float(r.abs().sum()) == 0
    assert s1.all_gather(x) is x
    for rank in range(2):
        rp = parallel.ReplayComm(rec.trace)
        s2 = parallel.FrameShard(T, rank, 2, rp)
        xl = x[rank * 2 * hw:(rank + 1) * 2 * hw]
        l, r = torch.empty(hw, C), torch.empty(hw, C)
        s2.halo(xl, hw, l, r)
        if rank == 0:
            assert float(l.abs().sum()) == 0 and torch.equal(r, x[2 * hw:3 * hw])
        else:
            assert torch.equal(l, x[hw:2 * hw]) and float(r.abs().sum()) == 0
        assert torch.equal(s2.all_gather(xl), x)
        assert rp.worst == 0.0 and rp.pos == 2
    with pytest.raises(ValueError):
        parallel.FrameShard(5, 0, 2)


# ----------------------------------------------------------------------------------------------------------------------
# tile sharding of aggregation sampling (parallel.TileShard, BASELINE configs[3]): the per-step exchange of the tiles' eps
# ----------------------------------------------------------------------------------------------------------------------
def _tile_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from mgld_vsr_amd import parallel
    parallel.init(backend="gloo")
    n_tiles, T, c, ts = 9, 2, 4, 6
    sh = parallel.TileShard(n_tiles, rank, world)
    # every tile's "eps" encodes its global tile index; missing tiles of the last ranks are zero padding
    loc = torch.zeros(sh.per * T, c, ts, ts)
    for j, k in enumerate(range(sh.k0, sh.k1)):
        loc[j * T:(j + 1) * T] = float(k + 1)
    buf = torch.full((world * sh.per * T, c, ts, ts), -1.0)
    full = sh.gather(loc, out=buf)
    assert full is buf
    got = [float(full[sh.slot(k) * T, 0, 0, 0]) for k in range(n_tiles)]
    ret[rank] = (got, sh.k0, sh.k1, tuple(full.shape))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_tile_shard_gather(world):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_tile_worker, args=(world, port, ret), nprocs=world, join=True)
    per = -(-9 // world)
    covered = []
    for r in range(world):
        got, k0, k1, shape = ret[r]
        assert got == [float(k + 1) for k in range(9)]        # every rank sees every tile, addressed by slot(k)
        assert shape == (world * per * 2, 4, 6, 6)
        covered += list(range(k0, k1))
    assert covered == list(range(9))                           # contiguous blocks, every tile exactly once


def test_tile_shard_replay_matches_recording():
    sys.path.insert(0, ROOT)
    from mgld_vsr_amd import parallel
    n_tiles, rows = 9, 3
    full = torch.randn(n_tiles * rows, 5)
    rec = parallel.RecordingComm()
    assert parallel.TileShard(n_tiles, 0, 1, rec).gather(full) is full
    for world in (2, 4):
        for rank in range(world):
            rp = parallel.ReplayComm(rec.trace)
            sh = parallel.TileShard(n_tiles, rank, world, rp)
            loc = torch.zeros(sh.per * rows, 5)
            loc[:(sh.k1 - sh.k0) * rows] = full[sh.k0 * rows:sh.k1 * rows]
            out = sh.gather(loc)
            assert out.shape[0] == world * sh.per * rows and rp.worst == 0.0
            for k in range(n_tiles):
                assert torch.equal(out[sh.slot(k) * rows:(sh.slot(k) + 1) * rows], full[k * rows:(k + 1) * rows])


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus N` (no torchrun around it, as the driver invokes it) must start the N ranks itself: the launcher is
    re-executed under torch.distributed.run and rank 0 reports the world size the process group really had (gloo here; the GPU
    path differs only in the backend)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--spawn-selftest"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["max_over_ranks_ok"] is True


@pytest.mark.parametrize("flags,mode,n", [([], "segment", 8), (["--frame-shard"], "frame", 8), (["--tile", "--tile-shard", "--size", "1024", "--frames", "4"], "tile", 4)])
def test_bench_launcher_and_comm_plan_of_every_multi_gpu_mode(flags, mode, n):
    """`bench.py --gpus 8` / `--frame-shard` / `--tile-shard` through the FULL launcher path (bench.py re-executes itself
    under torch.distributed.run, one process per rank, rendezvous on 127.0.0.1) with the exchanges of the mode replayed over the real
    process group (gloo here, RCCL on a multi-GPU lease): every rank completes, rank 0 reports the world size it saw and the bytes each
    rank sends per step / per segment (parallel.comm_plan: the figures DESIGN.md section 4 quotes)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--spawn-selftest"] + flags, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == n and res["world_size_seen"] == n and res["mode"] == mode and res["max_over_ranks_ok"] is True
    from mgld_vsr_amd import parallel
    if mode == "segment":
        assert res["comm_bytes_per_step_per_rank"] == 0 and res["comm_bytes_per_segment_per_rank"] == 0      # no data-path collective
    else:
        assert res["comm_bytes_per_step_per_rank"] > 0 and len(res["dry_run_bytes_by_rank"]) == n
        assert len(set(res["dry_run_bytes_by_rank"])) == 1                                                    # every rank ran the whole plan


def test_comm_plan_figures():
    """the bytes of the frame-sharded mode at the shipped geometry (8 x 512^2 over 8 ranks): 2 + 1 + 1 exchanges per step, 13 halo
    exchanges + 2 all-gathers per segment"""
    from mgld_vsr_amd import parallel
    p = parallel.comm_plan("frame", T=8, H=512, W=512, world=8, steps=50)
    assert [e[1] for e in p["per_step"]] == ["p2p", "all_gather", "all_gather"] and p["per_step"][0][2] == 2
    mid = 8 * 8 * 1280 * 2
    assert p["per_step"][0][3] == 2 * mid and p["per_step"][1][3] == 3 * mid and p["per_step"][2][3] == 4 * 64 * 64 * 4
    halos = [e for e in p["per_segment"] if e[1] == "p2p"]
    assert len(halos) == 5 and halos[-1][3] == 2 * 3 * 512 * 512 * 128 * 2            # 1 + 4 x 3 = 13 temporal convolutions
    assert p["bytes_per_segment"] == 50 * p["bytes_per_step"] + sum(e[2] * e[3] for e in p["per_segment"])
    assert parallel.comm_plan("segment", world=8)["bytes_per_segment"] == 0


# ----------------------------------------------------------------------------------------------------------------------
# the frame-split video decode at world 8: exchanged bytes LOGGED by DistComm.report() == comm_plan()'s prediction
# ----------------------------------------------------------------------------------------------------------------------
def _decode_comm_worker(rank, world, port, ret):
    """walks the REAL decoder module (VideoDecoder_Mix at the shipped width, meta parameters) the way its run() does — conv_in, mid
    block + temporal_mixing, per level the res-blocks each followed by a temporal_mixing, upsample — and performs the one-frame halo
    exchange of every SpatialTemporalConv with tensors of the real shape through FrameShard.halo / DistComm (gloo), one frame per rank"""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from mgld_vsr_amd import parallel
    from mgld_vsr_amd.pipeline import model_configs
    from mgld_vsr_amd.unet import SpatialTemporalConv
    from mgld_vsr_amd.vae import VideoDecoder_Mix
    parallel.init(backend="gloo")
    T, H = world, 64
    dd = dict(model_configs(T)[1]["params"]["ddconfig"], resolution=H)
    dec = VideoDecoder_Mix(**dd)
    sh = parallel.FrameShard(T, rank, world)
    parallel.DistComm.measure = True
    parallel.DistComm.report()
    n_tconv = 0

    def halo(mod, h, w):
        nonlocal n_tconv
        assert isinstance(mod, SpatialTemporalConv)
        C = mod.temporal_conv.weight.shape[0]
        x = torch.full((sh.F * h * w, C), float(rank), dtype=torch.float16)
        left, right = torch.empty(h * w, C, dtype=torch.float16), torch.empty(h * w, C, dtype=torch.float16)
        sh.halo(x, h * w, left, right)
        assert float(left[0, 0]) == (rank - 1 if rank > 0 else 0.0) and float(right[0, 0]) == (rank + 1 if rank < world - 1 else 0.0)
        n_tconv += 1
    h = w = H // 8
    halo(dec.temporal_mixing, h, w)
    for lvl in reversed(range(dec.num_resolutions)):
        for b in range(dec.num_res_blocks + 1):
            halo(dec.up[lvl].temporal_mixing[b], h, w)
        if lvl != 0:
            h, w = 2 * h, 2 * w
    rep = parallel.DistComm.report()
    parallel.DistComm.measure = False
    ret[rank] = (n_tconv, rep, dd["ch"], tuple(dd["ch_mult"]))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_frame_sharded_decode_comm_bytes_match_the_plan_at_world_8():
    """the halo bytes of the frame-split video decode, logged exchange by exchange at world 8 (gloo), equal what
    parallel.comm_plan() predicts for the same geometry: middle ranks send to both neighbours, the two end ranks to one"""
    sys.path.insert(0, ROOT)
    from mgld_vsr_amd import parallel
    world = 8
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_decode_comm_worker, args=(world, port, ret), nprocs=world, join=True)
    n_tconv, _, ch, ch_mult = ret[0]
    plan = parallel.comm_plan("frame", T=world, H=64, W=64, world=world, steps=50, ch=ch, ch_mult=ch_mult)
    halos = [(c, b) for what, kind, c, b in plan["per_segment"] if kind == "p2p"]
    assert n_tconv == 1 + sum(3 for _ in ch_mult) == 13
    want_mid = sum(c * b for c, b in halos)                 # both neighbours
    for r in range(world):
        nt, rep, _, _ = ret[r]
        assert nt == n_tconv and rep["calls"] == n_tconv
        assert rep["bytes"] == (want_mid if 0 < r < world - 1 else want_mid // 2), (r, rep, want_mid)
