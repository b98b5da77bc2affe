#!/usr/bin/env python
"""bench.py — HR frames/s of the MGLD-VSR per-segment hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic segment per GPU: VAE encode (x2, as the reference script
does: init latent + decoder features) -> 50-step respaced DDPM sampling over the struct-cond encoder + UNet ->
temporal-aware VAE sequence decode -> AdaIN colour fix.  Workload = BASELINE.json configs[1]: 8 frames at 512x512
(latent 64x64x4), 50 DDPM steps, random-init weights of the shipped architecture, flow warp off.  Inputs are resident
in HBM when the timed region starts.  N>1: one process per GPU (torch.distributed / RCCL), each rank owns its own
segment (segments are independent in the reference: no data-path collective) -> weak scaling.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
# the sources whose sha keys profiles/r06_pmc_traffic.json (tools/pmc_traffic.sh imports this list)
GEMM_FAMILY_SOURCES = ("igemm_common.h", "pp_common.h", "igemm.hip", "conv3q.hip", "conv3r.hip", "ppgemm.hip", "pptconv.hip", "attention.hip")
sys.path.insert(0, ROOT)

PEAK_FP16_TFLOPS = 2500.0   # dense MFMA fp16 peak, /opt/skills/guides/MI355X_MICROARCH.md
# algorithmic work per HR frame (measured with FlopCounterMode on the reference)
GFLOP_STEP_PER_FRAME = 967.4
GFLOP_ENC_PER_FRAME = 1116.7
GFLOP_DEC_PER_FRAME = 4139.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--ddpm-steps", type=int, default=50)
    ap.add_argument("--guidance", action="store_true", help="flow-guided latent warp on (configs[2]); default off (configs[1])")
    ap.add_argument("--raft", action="store_true", help="with --guidance: estimate the flows with RAFT_SR inside the timed step")
    ap.add_argument("--tile", action="store_true", help="aggregation sampling over 64x64 latent tiles, overlap 32 (configs[3]: use with --size 1024)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay (PMC profiling runs)")
    ap.add_argument("--frame-shard", action="store_true",
                    help="N>1: split the frames of ONE segment over the ranks (halo exchange / all-gather over RCCL, strong "
                         "scaling of a single segment) instead of one segment per rank")
    ap.add_argument("--tile-shard", action="store_true",
                    help="N>1 with --tile: split the latent tiles of ONE segment's aggregation sampling over the ranks (one all-gather "
                         "of the tiles' eps per step; BASELINE configs[3]) instead of one segment per rank")
    ap.add_argument("--inflight", type=int, default=0,
                    help="segments kept in flight on ONE GPU (one host thread + stream + pipeline instance each).  1 = the reference's "
                         "one-segment-at-a-time loop.  With 2-3 in flight one segment's kernels fill the tile-quantisation tails and the "
                         "latency-bound stretches of the others: same result per segment (tested bit for bit), higher frames/s, "
                         "proportionally longer per-segment latency.  0 (default) = 3 up to 8 x 512^2 frames per segment, 2 up to twice "
                         "that, else 1 (arena memory)")
    ap.add_argument("--clips", type=int, default=0,
                    help="independent segments batched as CLIPS of one pass (round 5): a step runs `clips` segments of --frames frames through "
                         "ONE encode / sampling / decode pass (the UNet sees clips x frames frames per launch, each clip with its own temporal "
                         "windows, flows and noise), so the low-resolution levels and the projections get clips x the rows per launch.  The "
                         "result of every clip is what it produces alone (tested).  0 (default) = 2 for segments up to 8 x 512^2 frames in the plain "
                         "segment-parallel mode (measured best with two such passes in flight: profiles/r05_clips_inflight.txt), else 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-one-at-a-time", action="store_true", help="skip the second scheduling leg (profiling runs of the segments in flight)")
    ap.add_argument("--small", action="store_true", help="reduced-width nets (plumbing check only; not a valid bench)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--spawn-selftest", action="store_true",
                    help="launcher check without a GPU: start the N ranks, rendezvous, barrier + max-reduce, print the JSON "
                         "skeleton with the world size the ranks saw (tests/test_parallel_cpu.py)")
    ap.add_argument("--dump-shapes", default=None, help="write the per-problem table of the roofline pass to this JSON file")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves (one process per GPU, the
    reference's own scheme: N processes, `--n_gpus N --select_idx r`, oldcanvas_tile.py:337-339) by re-executing this file
    under torch.distributed.run, rendezvous on 127.0.0.1, and pass rank 0's JSON line through."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def dist_setup(n, backend="nccl"):
    if n <= 1:
        return 0, 1, 0
    from mgld_vsr_amd import parallel
    rank, world, local = parallel.init(backend=backend)   # nccl = RCCL over xGMI; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from torchrun
    if world != n:
        raise SystemExit(f"bench.py --gpus {n}: the process group has {world} ranks")
    return rank, world, local


def build_pipeline(args):
    from mgld_vsr_amd import build
    build.build(verbose=False)
    from mgld_vsr_amd.pipeline import VSRPipeline, model_configs
    cfgs = None
    if args.small:
        cfgs = model_configs(args.frames, unet_overrides=dict(model_channels=64, context_dim=64, semb_channels=64),
                             struct_overrides=dict(model_channels=64, out_channels=64, num_heads=1),
                             vae_overrides=dict(ch=32), context_dim=64)
    else:
        cfgs = model_configs(args.frames)
    return VSRPipeline(num_frames=args.frames, ddpm_steps=args.ddpm_steps, configs=cfgs)


def make_inputs(pipe, args, rank):
    """inputs of one step: `clips` independent segments (own frames / noise / flows each) concatenated along the frame axis"""
    if args.clips > 1:
        one = argparse.Namespace(**dict(vars(args), clips=1))
        parts = [make_inputs(pipe, one, rank + 100003 * i) for i in range(args.clips)]
        frames = torch.cat([p[0] for p in parts])
        noise = {"posterior": torch.cat([p[1]["posterior"] for p in parts]), "x_T": torch.cat([p[1]["x_T"] for p in parts]),
                 "steps": torch.cat([p[1]["steps"] for p in parts], 1)}
        flows = masks = None
        if parts[0][2] is not None:
            flows = tuple(torch.cat([p[2][j] for p in parts]) for j in range(2))
            masks = tuple(torch.cat([p[3][j] for p in parts]) for j in range(2))
        return frames, noise, flows, masks
    from mgld_vsr_amd import synth
    T, S, H = args.frames, args.ddpm_steps, args.size
    dev = pipe.engine().device
    g = torch.Generator(device="cpu").manual_seed(1000 + rank)
    frames = (torch.rand(T, 3, H, H, generator=g) * 2 - 1).to(dev)            # U(-1,1) LR-upsampled frames
    h = H // 8
    noise = {"posterior": torch.randn(T, 4, h, h, generator=g).to(dev), "x_T": torch.randn(T, 4, h, h, generator=g).to(dev),
             "steps": torch.randn(S, T, 4, h, h, generator=g).to(dev)}
    flows = masks = None
    if args.guidance:
        from mgld_vsr_amd.flowops import forward_backward_consistency_check
        ff, fb = synth.smooth_flow("bench/ff", T - 1, h, h).to(dev), synth.smooth_flow("bench/fb", T - 1, h, h).to(dev)
        fo, bo = forward_backward_consistency_check(fb, ff)
        flows, masks = (ff[None], fb[None]), (fo[None, :, None], bo[None, :, None])
    return frames, noise, flows, masks


def igemm_algo_bytes(p):
    """algorithmic HBM bytes of one igemm launch: input activations once (not once per tap), weights, residual, output"""
    b = max(1, p.batch)
    if p.mode == 1:      # CONV3X3: frames * Hin * Win * Cin input elements
        a_elems = (p.M // max(1, p.Hout * p.Wout)) * p.Hin * p.Win * p.Cin
    elif p.mode == 2:    # TCONV3
        a_elems = p.M * p.Cin
    else:
        a_elems = p.M * p.K
    n_out = p.N // 2 if p.act == 4 else p.N
    out_b = 4 if p.out_f32 else 2
    return b * (2.0 * a_elems + 2.0 * p.N * p.K + out_b * p.M * n_out + (2.0 * p.M * n_out if p.R else 0.0))


HBM_PEAK_GBPS = 8000.0      # HBM3E spec peak (MI355X_MICROARCH.md; ~6300 GB/s achievable with a float4 copy)


def _pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate runs, FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for gfx950; tools/pmc_traffic.sh).  PMC counters cannot be read from inside this process;
    the file is only used when it was taken for THIS kernel name on THIS kernel source (sha256 of the GEMM family's and the attention kernel's source files),
    else null."""
    import hashlib
    try:
        src = b""
        for f in GEMM_FAMILY_SOURCES:
            with open(os.path.join(ROOT, "mgld_vsr_amd", "csrc", f), "rb") as fh:
                src += fh.read()
        sha = hashlib.sha256(src).hexdigest()[:16]
    except OSError:
        return None
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                pm = json.load(fh)
        except (OSError, ValueError):
            continue
        ent = pm.get("kernels", {}).get(kernel)
        if ent and pm.get("gemm_src_sha16") == sha:
            _pmc_traffic.source = f"profiles/{name} (rocprofv3 --pmc passes of this build, committed; not measured in this run)"
            return round(ent["hbm_bytes_per_launch"])
    return None


_pmc_traffic.source = None


ROCPROF_SUMMARY = "r06_kernel_stats.json"   # profiles/: rocprofv3 --kernel-trace --stats summary of `bench.py --clips 2 --inflight 1 --steps 1 --warmup 1` (tools/prof_run.sh + tools/kstats.py)


def _rocprof_summary():
    try:
        with open(os.path.join(ROOT, "profiles", ROCPROF_SUMMARY)) as fh:
            return json.load(fh).get("kernels", {})
    except (OSError, ValueError):
        return {}


def _rocprof_avg(kernel):
    """average launch duration (us) of `kernel` in the committed rocprofv3 --kernel-trace --stats summary of this round's bench
    command (profiles/ROCPROF_SUMMARY), or None"""
    for name, ent in _rocprof_summary().items():
        if kernel.replace(" ", "") in name.replace(" ", ""):
            return ent
    return None


def _rocprof_rank(names):
    """`names` (kernel instantiations timed in this run) ordered by their TOTAL time in the committed rocprofv3 summary, largest first; names the
    summary does not hold are dropped.  The roofline's dominant kernel is the first of these: the top row of the committed profile, not
    whichever of two co-dominant kernels happened to run 1 ms longer in this process."""
    ks = _rocprof_summary()
    tot = {}
    for n in names:
        for name, ent in ks.items():
            if n.replace(" ", "") in name.replace(" ", ""):
                tot[n] = tot.get(n, 0.0) + ent.get("total_ms", 0.0)
    return sorted(tot, key=lambda n: -tot[n])


def roofline(pipe, args, frames, noise, flows, masks):
    """One more pass of the same segment with EVERY launch of the GEMM family (implicit-GEMM / patch-conv kernels incl. their
    split-K reduce), the flash attention and the HBM-bound norm kernels bracketed by hipEvents on the launch stream, IN SEQUENCE
    (eager launches, the same launch list the hipGraph replays): each kernel is timed where it sits in the pipeline, with its
    operands in the cache state its producers left them.  Launches are grouped by the kernel instantiation the launcher picked
    (mgld_igemm_kernel_name: the name rocprofv3 prints), so `roofline` and profiles/r02_kernel_stats.txt describe the same kernels.
    `roofline` = the kernel with the largest total time."""
    from mgld_vsr_amd import hip
    # what a hipEvent pair costs by itself (two records back to back on the same stream, nothing between them): subtracted from
    # every bracketed launch, so that the 6-15 us norm kernels are not charged ~2 us of event processing each
    cal = [(hip.Event(), hip.Event()) for _ in range(64)]
    for a, b in cal:
        a.record()
        b.record()
    torch.cuda.synchronize()
    ev_ms = sorted(a.elapsed_ms(b) for a, b in cal)[len(cal) // 2]
    passes = []
    for _ in range(2):   # the launch list is deterministic: two passes, per-launch minimum (one stray stall of tens of ms in a
        hip.TIMED = []   # single pass would otherwise be charged to whichever kernel it hit)
        pipe.run_segment(frames, flows=flows, masks=masks, noise=noise, use_graph=False, tile=TILE)
        torch.cuda.synchronize()
        passes.append([(kind, info, e0.elapsed_ms(e1)) for kind, info, e0, e1 in hip.TIMED])
    hip.TIMED = None
    if not (len(passes[0]) == len(passes[1]) and all(a[0] == b[0] for a, b in zip(*passes))):
        passes[0] = passes[1]   # a pass that still did one-time work (cache fills) has another launch list: keep the steady one
    kern, shapes, hbm = {}, {}, {}
    tmin = {}
    for (kind, info, t0), (_, _, t1) in zip(*passes):
        # AVERAGE of the two passes (round 3 took the per-launch minimum, which read 7 % above rocprofv3's average for the dominant
        # kernel); a pass that caught a stall of > 3x the other one is dropped for that launch.  The minimum is reported beside it.
        lo, hi = min(t0, t1), max(t0, t1)
        ms = max((lo if hi > 3.0 * lo else 0.5 * (t0 + t1)) - ev_ms, 1e-4)
        ms_min = max(lo - ev_ms, 1e-4)
        if kind == "igemm":
            p = info
            name, splits = hip.igemm_kernel_name(p)
            k = kern.setdefault(name, {"flops": 0.0, "ms": 0.0, "launches": 0, "bytes": 0.0, "splitk_launches": 0})
            k["flops"] += hip.igemm_flops(p)
            k["bytes"] += igemm_algo_bytes(p)
            k["ms"] += ms
            tmin[name] = tmin.get(name, 0.0) + ms_min
            k["launches"] += 1
            k["splitk_launches"] += 1 if splits > 1 else 0
            key = (name, splits, p.mode, p.M, p.N, p.K, p.Cin, p.Hin, p.Win, p.stride, p.up2, p.act, max(1, p.batch))
            g = shapes.setdefault(key, {"count": 0, "ms": 0.0, "flops": hip.igemm_flops(p)})
            g["count"] += 1
            g["ms"] += ms
            if p.mode == 2 and p.M >= 65536:
                # temporal Conv3d (3,1,1) of the video decoder at its 128^2..512^2 levels: K = 3*C with C = 128..512 is 128-256 FLOP/B,
                # i.e. HBM-bound: graded on algorithmic bytes (each frame row read once, written once) over time
                h = hbm.setdefault("tconv_vae", {"bytes": 0.0, "ms": 0.0, "launches": 0})
                h["bytes"] += igemm_algo_bytes(p)
                h["ms"] += ms
                h["launches"] += 1
        elif kind == "attention":
            name = info["kernel"]      # as rocprofv3 prints it (mgld_attention_kernel_name)
            k = kern.setdefault(name, {"flops": 0.0, "ms": 0.0, "launches": 0, "bytes": 0.0, "splitk_launches": 0})
            k["flops"] += info["flops"]
            k["bytes"] += info["bytes"]
            k["ms"] += ms
            tmin[name] = tmin.get(name, 0.0) + ms_min
            k["launches"] += 1
        else:
            h = hbm.setdefault(kind, {"bytes": 0.0, "ms": 0.0, "launches": 0})
            h["bytes"] += info["bytes"]
            h["ms"] += ms
            h["launches"] += 1
    rows = [{"kernel": k[0], "splits": k[1], "mode": k[2], "M": k[3], "N": k[4], "K": k[5], "Cin": k[6], "H": k[7], "W": k[8], "stride": k[9],
             "up2": k[10], "act": k[11], "batch": k[12], "count": g["count"], "us": round(1e3 * g["ms"] / g["count"], 2),
             "tflops": round(g["flops"] * g["count"] / (g["ms"] * 1e-3) / 1e12, 1), "total_ms": round(g["ms"], 2)} for k, g in shapes.items()]
    rows.sort(key=lambda r: -r["total_ms"])
    dump = args.dump_shapes or (os.path.join(ROOT, "gpurun_out", "igemm_shapes.json") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None)
    if dump:   # per-problem table (kernel tuning; a copy of the full-width run is committed under profiles/)
        with open(dump, "w") as fh:
            json.dump(rows, fh, indent=0)
    ranked = _rocprof_rank(list(kern))                      # deterministic: the order of the committed rocprofv3 summary
    by_time = sorted(kern, key=lambda n: -kern[n]["ms"])    # fallback (no summary of this build's kernels): this run's in-sequence times
    order = ranked + [n for n in by_time if n not in ranked]
    dom, second_name = order[0], (order[1] if len(order) > 1 else None)
    d = kern[dom]
    achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
    gemm = [v for n, v in kern.items() if not n.startswith("flash_attn")]
    all_flops, all_ms = sum(v["flops"] for v in gemm), sum(v["ms"] for v in gemm)
    def _entry(n, v):
        e = {"kernel": n, "ms_per_segment": round(v["ms"], 2), "launches": v["launches"], "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1),
             "frac": round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / PEAK_FP16_TFLOPS, 4), "algorithmic_bytes": round(v["bytes"] / v["launches"])}
        tr = _pmc_traffic(n)          # HBM-side bytes per launch from the committed PMC pass, when it was taken on this build
        if tr:
            e["traffic"], e["traffic_over_algorithmic"] = tr, round(tr / max(1.0, v["bytes"] / v["launches"]), 2)
        rk = _rocprof_avg(n)          # the same kernel's average launch in the committed rocprofv3 --kernel-trace summary
        if rk:
            e["frac_rocprof_avg"] = round(v["flops"] / v["launches"] / (rk["avg_us"] * 1e-6) / 1e12 / PEAK_FP16_TFLOPS, 4)
        return e
    by_kernel = [_entry(n, v) for n, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms"])[:10]]
    traffic = _pmc_traffic(dom)
    rp = _rocprof_avg(dom)
    hbm_out = {n: {"achieved_gbps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1), "frac_of_peak": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                   "ms_per_segment": round(v["ms"], 2), "launches": v["launches"], "avg_launch_us": round(1e3 * v["ms"] / v["launches"], 2)}
               for n, v in hbm.items()}
    return {
        # the pass that was timed: `clips` segments batched as clips of ONE pass (the default scheduling's launch shapes); every
        # "..._per_segment" figure below is per such PASS, launches likewise — frac / achieved do not depend on that
        "segments_per_pass": int(args.clips),
        "bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": PEAK_FP16_TFLOPS, "unit": "TFLOP/s",
        "frac": round(achieved / PEAK_FP16_TFLOPS, 4),
        # the same with the per-launch MINIMUM of the two passes (what round 3 reported), and with the average launch duration of the
        # committed rocprofv3 --kernel-trace --stats summary of this command (profiles/ROCPROF_SUMMARY) when it has this kernel
        "frac_event_min": round(d["flops"] / (tmin[dom] * 1e-3) / 1e12 / PEAK_FP16_TFLOPS, 4),
        "frac_rocprof_avg": (round(d["flops"] / d["launches"] / (rp["avg_us"] * 1e-6) / 1e12 / PEAK_FP16_TFLOPS, 4) if rp else None),
        "rocprof_avg_us": (rp["avg_us"] if rp else None),
        "traffic": traffic, "traffic_source": _pmc_traffic.source if traffic else None,
        "algorithmic_bytes": round(d["bytes"] / d["launches"]),   # per launch: every operand element moved once
        "launches_per_segment": d["launches"], "splitk_launches": d["splitk_launches"], "avg_launch_us": round(1e3 * d["ms"] / d["launches"], 2),
        "kernel_ms_per_segment": round(d["ms"], 2),
        "timing": "hipEvents around every launch, in sequence (two eager passes of the same launch list, per-launch AVERAGE); empty-bracket cost subtracted",
        "event_pair_us": round(1e3 * ev_ms, 2),
        "all_gemm": {"tflops": round(all_flops / (all_ms * 1e-3) / 1e12, 2), "frac": round(all_flops / (all_ms * 1e-3) / 1e12 / PEAK_FP16_TFLOPS, 4),
                     "ms_per_segment": round(all_ms, 2), "gflop_per_segment": round(all_flops / 1e9, 1)},
        "kernel_pick": (f"top row of profiles/{ROCPROF_SUMMARY} (committed rocprofv3 --kernel-trace --stats summary of this command)" if ranked and ranked[0] == dom
                        else "largest in-sequence time of this run (the committed rocprofv3 summary does not hold this build's kernels)"),
        # the co-dominant kernel beside it (second row of the same summary): the two differ by a few ms per pass and by 0.1 in `frac`
        "second": (_entry(second_name, kern[second_name]) if second_name else None),
        "by_kernel": by_kernel,
        "hbm": {"peak_gbps": HBM_PEAK_GBPS, "kernels": hbm_out},   # the HBM-bound list of SURVEY 8(d): achieved = algorithmic bytes / time
    }


def cpu_baseline(args):
    """The oracle (CPU restatement, fp32, up to 32 torch threads) timed on this box's host cores, on a bounded sample of the workload
    (configs[0] run fully, configs[1] extrapolated from >= 2 measured steps):
      * BASELINE configs[0] in full — one 512x512 frame, 4 DDPM steps (struct-cond + UNet per step), two VAE encodes, the video decode;
      * BASELINE configs[1]'s OWN batch — the 8-frame 512x512 clip the GPU line is quoted on: two DDPM steps (struct-cond + UNet on all
        eight frames, temporal modules over T = 8), one VAE encode and one video decode of the eight frames.
    `value` = HR frames/s of the 50-step pipeline from the 8-frame clip's measured mean step time (every step is identical work)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from configs import STRUCT_FULL, UNET_FULL, VAE_DD_FULL
    from mgld_vsr_amd import synth
    from oracle import nets as onets
    cores = min(32, os.cpu_count() or 1)   # more threads than this thrash on the small per-layer problems
    torch.set_num_threads(cores)
    from ldm.models.autoencoder import VideoAutoencoderKLResi
    from ldm.modules.diffusionmodules.openaimodel import InflatedEncoderUNetModelWT, InflatedUNetModelDualcondV2

    def names(m):
        return [(k, tuple(v.shape)) for k, v in m.state_dict().items() if v.is_floating_point()]
    h = args.size // 8
    Tb = int(args.frames)                  # configs[1]: 8
    ctx = synth.synth_tensor("ctx", (1, 77, 1024))
    out = {}
    for Tc, nsteps in ((1, 4), (Tb, 2)):
        ucfg, scfg, vdd = dict(UNET_FULL, num_frames=Tc), dict(STRUCT_FULL, num_frames=Tc), dict(VAE_DD_FULL, num_frames=Tc)
        usd = synth.synth_state_dict(names(InflatedUNetModelDualcondV2(**ucfg)), "unet")
        ssd = synth.synth_state_dict(names(InflatedEncoderUNetModelWT(**scfg)), "structcond")
        vsd = synth.synth_state_dict(names(VideoAutoencoderKLResi(ddconfig=vdd, lossconfig={"target": "torch.nn.Identity"},
                                                                  embed_dim=4)), "vae")
        x, lat = synth.synth_tensor("cpu/x", (Tc, 4, h, h)), synth.synth_tensor("cpu/lat", (Tc, 4, h, h), 0.5)
        img = synth.synth_tensor("cpu/img", (Tc, 3, args.size, args.size), 0.5)
        t_steps = []
        with torch.no_grad():
            for k in range(nsteps):
                t = torch.tensor([999 - 250 * k] * Tc)
                t0 = time.time()
                sc = onets.structcond_forward(ssd, scfg, lat, t)
                eps = onets.unet_forward(usd, ucfg, x, t, ctx, sc)
                t_steps.append(time.time() - t0)
                x = x - 0.1 * eps                  # (a stand-in for the posterior step: keeps consecutive steps on different data)
            t0 = time.time()
            _, _, fea = onets.vae_moments(vsd, vdd, img)
            t_enc = time.time() - t0
            t0 = time.time()
            onets.vae_decode(vsd, vdd, x, fea)
            t_dec = time.time() - t0
        out[Tc] = (t_steps, t_enc, t_dec)
        del usd, ssd, vsd, fea
    (s1, e1, d1), (s2, e2, d2) = out[1], out[Tb]
    c0_total = sum(s1) + 2 * e1 + d1               # configs[0]: 4 steps + 2 encodes + decode of one frame
    per_frame = (args.ddpm_steps * (sum(s2) / len(s2)) + 2 * e2 + d2) / Tb
    return {"value": round(1.0 / per_frame, 5), "unit": "HR frames/s", "cores": cores, "kind": "port",
            "configs0": {"seconds": round(c0_total, 2), "hr_frames_per_s": round(1.0 / c0_total, 5), "step_seconds": [round(v, 2) for v in s1],
                         "what": "BASELINE configs[0] in full: one 512x512 frame, 4 DDPM steps, 2 VAE encodes, video decode"},
            "configs1_sample_seconds": round(sum(s2) + e2 + d2, 2),
            "note": "the CPU RESTATEMENT (oracle/, torch fp32 kernels) of the reference's algorithm, not the reference's own Python (which does "
                    "not travel to the GPU box; SURVEY.md quotes 0.0041 frames/s for it on other host cores)",
            "sample": f"oracle fp32: configs[0] in full ({c0_total:.1f}s: steps {', '.join(f'{v:.2f}' for v in s1)}s, encode {e1:.2f}s, decode {d1:.2f}s) + "
                      f"configs[1]'s own batch, the {Tb}-frame {args.size}x{args.size} clip: 2 DDPM steps {s2[0]:.2f}s / {s2[1]:.2f}s, 1 VAE encode of the "
                      f"{Tb} frames {e2:.2f}s, 1 video decode {d2:.2f}s; `value` = that clip's mean step x {args.ddpm_steps} + 2 encodes + 1 decode, per frame"}


TILE = None
GRAPH = True


def main():
    global TILE, GRAPH
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))      # plain `python bench.py --gpus N`: launch the N ranks, rank 0 prints the line
    TILE = (64, 32) if args.tile else None
    GRAPH = not args.no_graph
    if args.clips <= 0:     # default scheduling (measured, profiles/r05_clips_inflight.txt): two segments per pass, two passes in flight
        small = args.frames * args.size * args.size <= 8 * 512 * 512
        args.clips = 2 if (small and not (args.tile or args.frame_shard or args.tile_shard or args.spawn_selftest) and args.steps >= 2) else 1
    if args.spawn_selftest:
        from mgld_vsr_amd import parallel
        backend = "gloo" if not torch.cuda.is_available() else args.backend
        rank, world, local = dist_setup(args.gpus, backend)
        if backend == "nccl":
            torch.cuda.set_device(local)
        parallel.barrier(sync_device=False)
        dt = parallel.max_over_ranks(1e-3 * (rank + 1))
        # the exchanges of the selected multi-GPU mode, replayed with dummy tensors through the same DistComm calls (launcher,
        # rendezvous and transport check: gloo on CPU here, RCCL when a multi-GPU lease exists)
        mode = "frame" if args.frame_shard else ("tile" if args.tile_shard else "segment")
        plan = parallel.comm_plan(mode, T=args.frames, H=args.size, W=args.size, world=world, steps=args.ddpm_steps)
        sent = parallel.comm_dry_run(plan, device="cuda" if backend == "nccl" else "cpu")
        per_rank = parallel.gather_floats(float(sent))
        if rank == 0:
            print(json.dumps({"metric": "HR frames/sec at 512^2, 50 DDPM steps", "value": None, "n_gpus": world, "selftest": True,
                              "world_size_seen": world, "backend": backend, "mode": mode,
                              "max_over_ranks_ok": abs(dt - 1e-3 * world) < 1e-9,
                              "comm_bytes_per_step_per_rank": plan["bytes_per_step"], "comm_bytes_per_segment_per_rank": plan["bytes_per_segment"],
                              "dry_run_bytes_by_rank": per_rank}), flush=True)
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()
        return
    rank, world, local = dist_setup(args.gpus, args.backend)
    if world > 1:      # N ranks build their (synthetic) weights at the same time: share the host cores instead of oversubscribing them
        torch.set_num_threads(max(1, (os.cpu_count() or world) // world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback on the product path)")
    torch.cuda.set_device(local)
    pipe = build_pipeline(args)
    # the other instances of the segments-in-flight pool share this one's host weights (taken before its first launch)
    spare = [pipe.clone_shared() for _ in range(2)] if (args.inflight == 0 or args.inflight > 1) and not (args.frame_shard or args.tile_shard) else []
    from mgld_vsr_amd import parallel
    shard = None
    if args.frame_shard and world > 1:
        # the frames of ONE segment over all ranks (second scheme): every rank builds the SAME clip
        frames, noise, flows, masks = make_inputs(pipe, args, 0)
        shard = parallel.FrameShard(args.frames, rank, world)
    else:
        frames, noise, flows, masks = make_inputs(pipe, args, rank)
        # every rank owns its own segment (weak scaling): the reference's `seq_idx % n_gpus == select_idx` sharding
        assert parallel.shard_segments(world, rank, world) == [rank]
    kw = dict(flows=flows, masks=masks, noise=noise, tile=TILE, use_graph=GRAPH)
    if shard is not None:
        kw.update(shard=shard, gather=True)
    if world > 1 and (args.frame_shard or args.tile_shard):
        parallel.DistComm.measure = True        # payload + device time of every exchange, reported next to comm_plan()'s prediction
    if args.tile_shard and args.tile and world > 1:
        frames, noise, flows, masks = make_inputs(pipe, args, 0)     # every rank builds the SAME clip
        h8 = args.size // 8
        n_tiles = len(pipe.model._tile_origins(h8, h8, TILE[0], TILE[1]))
        shard = parallel.TileShard(n_tiles, rank, world)
        kw = dict(flows=flows, masks=masks, noise=noise, tile=TILE, use_graph=GRAPH, tile_shard=shard)

    def own_flows(p, fr):        # --raft: the flows of every clip from its own frames (one RAFT batch per clip, as the script does per segment)
        est = [p.estimate_flows(fr[i * args.frames:(i + 1) * args.frames]) for i in range(args.clips)]
        return tuple(torch.cat([e[0][j] for e in est]) for j in range(2)), tuple(torch.cat([e[1][j] for e in est]) for j in range(2))

    def step():
        if args.raft and args.guidance:
            kw["flows"], kw["masks"] = own_flows(pipe, frames)
        return pipe.run_segment(frames, **kw)

    if args.inflight > 0:
        inflight = args.inflight
    else:
        px = args.clips * args.frames * args.size * args.size
        inflight = 3 if px <= 8 * 512 * 512 else (2 if px <= 16 * 512 * 512 else 1)
        if args.clips > 1:
            inflight = 2 if px <= 16 * 512 * 512 else 1
    if shard is not None:
        inflight = 1                          # the sharded modes spread ONE segment over the ranks
    inflight = max(1, min(inflight, args.steps))
    if inflight > 1:
        # K segments as `inflight` concurrent streams of K / inflight segments: every worker thread owns a pipeline instance (its own
        # engine, arena, hipGraph), a stream and a split-K scratch (the library keeps that per host thread); weights are the same
        # synthetic ones in every instance, every segment's result is what the one-at-a-time loop produces
        from mgld_vsr_amd.pipeline import SegmentPool
        pool = SegmentPool(lambda: build_pipeline(args), inflight, first=pipe, others=spare)
        del spare[:]
        ins = [(frames, noise, flows, masks)] + [make_inputs(pool.pipes[i], args, rank * inflight + i) for i in range(1, inflight)]

        def seg(pipe_i, j):                        # segment j on the instance that owns its inputs (j % inflight); --raft: the flows
            fr, nz, fl, mk = ins[j % inflight]     # are estimated inside the segment, as in step()
            if args.raft and args.guidance:
                fl, mk = own_flows(pipe_i, fr)
            return pipe_i.run_segment(fr, flows=fl, masks=mk, noise=nz, tile=TILE, use_graph=GRAPH)
        for i in range(inflight):                 # warm-up one instance at a time, on ITS inputs
            pool.map_on(i, seg, [i] * args.warmup)
        parallel.barrier()
        t0 = time.perf_counter()
        outs = pool.map(seg, list(range(args.steps)))
        parallel.barrier()
        dt_local = time.perf_counter() - t0
        dt = parallel.max_over_ranks(dt_local)
        out = torch.cat([o.float().reshape(-1)[:1024] for o in outs[-inflight:]])
        # MEASURED per-segment latency under this scheduling: a hipEvent pair around every segment on its worker's stream
        lat = sorted(pool.last_latency_ms.values())
        latency = {"median_ms": round(lat[len(lat) // 2], 1), "max_ms": round(lat[-1], 1), "min_ms": round(lat[0], 1),
                   "how": "hipEvent pair around each segment on its own stream, all timed segments"}
        # the OTHER scheduling, outside the contract's timed region: the same K segments one at a time on this rank's first
        # instance (the reference's loop); reported next to `value`, never instead of it
        # (with clips > 1 this leg runs ONE clip per pass: single segments, as the reference's loop does)
        T1 = args.frames
        one = (frames[:T1], {"posterior": noise["posterior"][:T1], "x_T": noise["x_T"][:T1], "steps": noise["steps"][:, :T1]},
               None if flows is None else tuple(f[:1] for f in flows), None if masks is None else tuple(m_[:1] for m_ in masks))

        def step1():
            fl, mk = one[2], one[3]
            if args.raft and args.guidance:
                fl, mk = pipe.estimate_flows(one[0])
            return pipe.run_segment(one[0], flows=fl, masks=mk, noise=one[1], tile=TILE, use_graph=GRAPH)
        if not args.no_one_at_a_time:
            step1()                                 # (the single-clip launch list has its own graph: capture it outside the timed leg)
        parallel.barrier()
        t1 = time.perf_counter()
        e_lat = []
        for j in range(0 if args.no_one_at_a_time else args.steps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            step1()
            b.record()
            e_lat.append((a, b))
        parallel.barrier()
        dt1 = parallel.max_over_ranks(time.perf_counter() - t1)
        l1 = sorted(a.elapsed_time(b) for a, b in e_lat)
        pool.close()
        one_at_a_time = None if args.no_one_at_a_time else {"value": round((1 if shard is not None else world) * args.frames * args.steps / dt1, 4), "ms_per_step": round(1e3 * dt1 / args.steps, 2),
                         "segment_latency_ms": round(l1[len(l1) // 2], 1), "steps": args.steps}
    else:
        for _ in range(args.warmup):
            step()
        parallel.barrier()                       # barrier + torch.cuda.synchronize() on both sides of the timed region
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        parallel.barrier()
        dt_local = time.perf_counter() - t0
        dt = parallel.max_over_ranks(dt_local)
        latency = {"median_ms": round(1e3 * dt / args.steps, 1), "how": "one segment at a time: wall time of the timed region / segments"}
        one_at_a_time = None
    comm = None
    if world > 1 and parallel.DistComm.measure:
        # warm-up + timed segments were logged alike: per segment = totals / (warmup + steps)
        rep, nseg = parallel.DistComm.report(), args.warmup + args.steps
        plan = parallel.comm_plan("tile" if args.tile_shard else "frame", T=args.frames, H=args.size, W=args.size, world=world, steps=args.ddpm_steps)
        comm = {"plan_bytes_per_segment_per_rank": plan["bytes_per_segment"], "measured_bytes_per_segment_rank0": int(rep["bytes"] / nseg),
                "measured_exchange_ms_per_segment_rank0": round(rep["ms"] / nseg, 3), "exchanges_per_segment": rep["calls"] // nseg,
                "how": "device event pair around every exchange on rank 0's stream (includes waiting for the slowest neighbour)"}
    ok = bool(torch.isfinite(out).all())
    h8_ = args.size // 8
    n_unet_tiles = len(pipe.model._tile_origins(h8_, h8_, TILE[0], TILE[1])) if TILE else (args.size / 512.0) ** 2
    per_rank_ms = parallel.gather_floats(1e3 * dt_local / args.steps) if world > 1 else [round(1e3 * dt_local / args.steps, 2)]
    ms_per_step = 1e3 * dt / args.steps
    segs = 1 if shard is not None else world
    fps = segs * args.clips * args.frames * args.steps / dt
    res = {
        "metric": "HR frames/sec at 512^2, 50 DDPM steps", "value": round(fps, 4), "unit": "HR frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2), "higher_is_better": True,
        "scaling": "strong" if shard is not None else "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"{args.frames}-frame {args.size}x{args.size} sequence (latent {args.size // 8}x{args.size // 8}x4), "
                               f"{args.ddpm_steps} DDPM steps, random-init SD-2.1 UNet + struct-cond encoder + KL-VAE encode x2 "
                               f"+ temporal video decoder + AdaIN, flow-guided warp {'on' if args.guidance else 'off'}{', aggregation sampling 64/32' if args.tile else ''}; "
                               + ("one segment, frames sharded over the GPUs" if shard is not None else
                                  ("one segment per GPU at a time" if inflight == 1 else f"independent segments, {inflight} in flight per GPU")
                                  + (f", {args.clips} segments batched as clips of each pass" if args.clips > 1 else "")),
                   "frames_per_segment": args.frames, "clips_per_pass": args.clips, "frames_per_step": args.clips * args.frames,
                   # what `value` presumes of the caller: this many independent segments pending per GPU; a single short
                   # video (one pending segment) runs at `value_one_at_a_time` / `one_at_a_time.segment_latency_ms`
                   "pending_segments_assumed_per_gpu": (1 if shard is not None else inflight * args.clips),
                   "parallelism": (f"tile-sharded x{world}" if args.tile_shard and shard is not None else f"frame-sharded x{world}")
                   if shard is not None else f"segment-parallel x{world}" + (f", {inflight} segments in flight per GPU" if inflight > 1 else ""),
                   "finite": ok,
                   "reduced_width": bool(args.small),
                   # how a sampling step is launched: one hipGraph, or (sharded modes) graph pieces around the collectives
                   "graphs_per_step": int(getattr(pipe.model, "last_graph_pieces", 0)),
                   "segments_in_flight": inflight,
                   "segment_latency_ms": latency["median_ms"], "segment_latency": latency,
                   "world_size_seen": world, "backend": args.backend if world > 1 else None, "per_rank_ms_per_step": per_rank_ms,
                   "comm": comm},
        # per frame: the sampler's work scales with the latent tiles it evaluates (aggregation sampling: every 64x64 tile is one 512^2
        # frame's worth of UNet + struct-cond work), the VAE's with the pixels
        "sustained_tflops": round(segs * args.clips * args.frames * (args.ddpm_steps * GFLOP_STEP_PER_FRAME * n_unet_tiles + (2 * GFLOP_ENC_PER_FRAME +
                                                          GFLOP_DEC_PER_FRAME) * (args.size / 512.0) ** 2) / 1e3 / (dt / args.steps), 1),
    }
    # whole-segment algorithmic FLOP rate against the dense fp16 MFMA peak of the GPUs in use (the path is compute-bound:
    # ~55 TFLOP per HR frame against ~0.15 TB of algorithmic HBM traffic)
    res["sustained_frac_of_mfma_peak"] = round(res["sustained_tflops"] / (PEAK_FP16_TFLOPS * world), 4)
    # the arithmetic this number was measured with, and what its precision features cost ("price it in the bench line"):
    # measured back to back on one box, profiles/r06_stream_lo.md / r06_ln_fold.md / r05_hp_encoder_kstats.txt
    eng_ = pipe.model.engine()
    res["arithmetic"] = {
        "operands": "fp16, fp32 MFMA accumulation",
        "residual_stream_two_fp16_planes": ",".join(sorted(eng_.lo_scopes)) or "off",
        "layernorm_folded_into_consumer": bool(eng_.LN_FOLD),
        "first_stage_encoder": "fp32 activations, split-fp16 contractions" if os.environ.get("MGLD_HP_ENCODER", "1") != "0" else "fp16",
        "weight_residual_pass": ",".join(sorted(eng_.w2_scopes)) or "off",
        "price": "two-plane stream: -4 % frames/s for -18 % latent / -28 % frame error (MGLD_STREAM_LO=0 MGLD_LN_FOLD=0: 14.25 against 13.63 / 13.89 against 13.33 on two boxes); "
                 "high-precision first-stage encoder: +29 ms per 8-frame encode (-2.5 %)",
    }
    if one_at_a_time is not None:     # both schedulings in one line: `value` = the default (segments in flight), this = the reference's loop
        res["value_one_at_a_time"] = one_at_a_time["value"]
        res["one_at_a_time"] = one_at_a_time
    else:
        res["value_one_at_a_time"] = res["value"]
    if rank == 0:
        if not args.no_roofline:
            res["roofline"] = roofline(pipe, args, frames, noise, flows, masks)
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(res), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
