"""Per-segment inference pipeline: the body of the reference's inference loop
(scripts/vsr_val_ddpm_text_T_vqganfin_oldcanvas_tile.py:296-329 model/schedule setup, :375-475 per-segment work)
as a reusable object.  Everything between "LR segment resident in HBM" and "HR frames resident in HBM" runs as
libmgld_hip launches on one stream.
"""
import copy
import os
import time

import numpy as np
import torch

from . import synth
from .ddpm import space_timesteps
from .flowops import adaptive_instance_normalization, wavelet_reconstruction
from .util import instantiate_from_config, load_trusted_checkpoint


def model_configs(num_frames=5, unet_overrides=None, struct_overrides=None, vae_overrides=None, context_dim=1024):
    """The `model:` sections of configs/mgldvsr/mgldvsr_512_realbasicvsr_deg.yaml:2-120 and
    configs/video_vae/video_autoencoder_kl_64x64x4_resi.yaml:1-54 as plain dicts (same targets / params)."""
    dd = dict(double_z=True, z_channels=4, resolution=512, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
              num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    dd.update(vae_overrides or {})
    unet = dict(num_frames=num_frames, image_size=32, in_channels=4, out_channels=4, model_channels=320,
                attention_resolutions=[4, 2, 1], num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_head_channels=64,
                use_spatial_transformer=True, use_linear_in_transformer=True, transformer_depth=1, context_dim=context_dim,
                use_checkpoint=False, legacy=False, semb_channels=256)
    unet.update(unet_overrides or {})
    struct = dict(num_frames=num_frames, image_size=96, in_channels=4, model_channels=256, out_channels=256,
                  num_res_blocks=2, attention_resolutions=[4, 2, 1], dropout=0, channel_mult=[1, 1, 2, 2], conv_resample=True,
                  dims=2, use_checkpoint=False, use_fp16=False, num_heads=4, num_head_channels=-1, num_heads_upsample=-1,
                  use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=False)
    struct.update(struct_overrides or {})
    diffusion = {
        "target": "ldm.models.diffusion.ddpm.LatentDiffusionVSRTextWT",
        "params": dict(
            linear_start=0.00085, linear_end=0.0120, num_timesteps_cond=1, log_every_t=200, timesteps=1000,
            first_stage_key="image", cond_stage_key="caption", image_size=512, channels=4, num_frames=num_frames,
            cond_stage_trainable=False, conditioning_key="crossattn", monitor="val/loss_simple_ema", scale_factor=0.18215,
            use_ema=False, train_temporal_module=True, unfrozen_diff=False, random_size=False, time_replace=1000, use_usm=True,
            unet_config={"target": "ldm.modules.diffusionmodules.openaimodel.InflatedUNetModelDualcondV2", "params": unet},
            first_stage_config={"target": "ldm.models.autoencoder.AutoencoderKL",
                                "params": {"embed_dim": 4, "monitor": "val/rec_loss", "ddconfig": dict(dd),
                                           "lossconfig": {"target": "torch.nn.Identity"}}},
            cond_stage_config={"target": "ldm.modules.encoders.modules.FrozenOpenCLIPEmbedder",
                               "params": {"freeze": True, "layer": "penultimate", "device": "cuda", "context_dim": context_dim}},
            structcond_stage_config={"target": "ldm.modules.diffusionmodules.openaimodel.InflatedEncoderUNetModelWT",
                                     "params": struct},
            flownet_config={"target": "basicsr.archs.raft_arch.RAFT_SR", "params": {"model": "normal", "load_path": None}}),
    }
    vae = {
        "target": "ldm.models.autoencoder.VideoAutoencoderKLResi",
        "params": dict(monitor="val/rec_loss", embed_dim=4, fusion_w=1.0, freeze_dec=True, synthesis_data=False, version=1,
                       lossconfig={"target": "torch.nn.Identity"}, ddconfig=dict(dd, num_frames=num_frames)),
    }
    return diffusion, vae


class VSRPipeline:
    def __init__(self, num_frames=5, ddpm_steps=50, dec_w=1.0, colorfix_type="adain", synthetic_weights=True, configs=None,
                 chunk_bytes=4 << 30):
        dcfg, vcfg = configs or model_configs(num_frames)
        self.model = instantiate_from_config(dcfg)
        self.vq_model = instantiate_from_config(vcfg)
        self.vq_model.decoder.fusion_w = dec_w
        self.num_frames, self.ddpm_steps, self.colorfix_type = num_frames, ddpm_steps, colorfix_type
        if synthetic_weights:
            synth.fill_module_(self.model.model.diffusion_model, "unet")
            synth.fill_module_(self.model.structcond_stage_model, "structcond")
            synth.fill_module_(self.model.first_stage_model, "first_stage")
            synth.fill_module_(self.vq_model, "vae")
            if hasattr(self.model.flownet_model, "fnet"):
                synth.fill_module_(self.model.flownet_model, "raft")
        self._setup_schedule(ddpm_steps)
        self.chunk_bytes = chunk_bytes

    def _setup_schedule(self, steps):
        """oldcanvas_tile.py:308-329"""
        m = self.model
        m.register_schedule(given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=0.00085,
                            linear_end=0.0120, cosine_s=8e-3)
        m.num_timesteps = 1000
        self.sqrt_alphas_cumprod = copy.deepcopy(m.sqrt_alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = copy.deepcopy(m.sqrt_one_minus_alphas_cumprod)
        self._sa999, self._soma999 = float(self.sqrt_alphas_cumprod[999]), float(self.sqrt_one_minus_alphas_cumprod[999])   # t = 999 (:449)
        use = set(space_timesteps(1000, [steps]))
        last, nb = 1.0, []
        for i, ac in enumerate(m.alphas_cumprod):
            if i in use:
                nb.append(1 - ac / last)
                last = ac
        m.register_schedule(given_betas=np.array([b.data.cpu().numpy() for b in nb]), timesteps=len(nb))
        m.num_timesteps = 1000
        m.ori_timesteps = sorted(list(use))

    def load_weights(self, ckpt=None, vqgan_ckpt=None):
        """the checkpoint handling both entry scripts share (oldcanvas_tile.py:91-108 / old.py): diffusion checkpoint, then the video
        VAE's.  A real diffusion checkpoint WITHOUT a video-VAE checkpoint is refused: the VAE's parameters would still be the zeros
        _finish_init left (synthetic weights are off) and the frames garbage without any error."""
        if ckpt:
            self.load_checkpoint(ckpt)
        if vqgan_ckpt:
            self.vq_model.init_from_ckpt(vqgan_ckpt)
        elif ckpt:
            raise SystemExit("[mgld] --ckpt was given but no --vqgan_ckpt exists: the video VAE would run with uninitialised weights; "
                             "pass --vqgan_ckpt (or drop --ckpt to run everything on the built-in synthetic weights)")
        return self

    def load_checkpoint(self, ckpt, vqgan_ckpt=None, context=None, verbose=True):
        """Load the reference's checkpoints the way its script does (oldcanvas_tile.py:91-108, 296-329): the diffusion model's
        `state_dict` goes, strict=False, into the model WITH ITS 1000-STEP SCHEDULE (the checkpoint stores betas / alphas_cumprod /
        posterior_* at length 1000; strict=False does not forgive size mismatches), and only then is the schedule respaced to
        `ddpm_steps`.  The text tower: when the checkpoint carries `cond_stage_model.model.*` (OpenCLIP text transformer) the tower
        is built to its shapes so those weights are USED; without them a precomputed empty-prompt embedding must be given
        (`context` [1,77,1024]) — a real checkpoint is never paired with the synthetic context silently.
        Returns (missing_keys, unexpected_keys) of the diffusion model."""
        m = self.model
        sd = load_trusted_checkpoint(ckpt) if isinstance(ckpt, str) else ckpt
        sd = sd["state_dict"] if "state_dict" in sd else sd
        cs = m.cond_stage_model
        tk = "cond_stage_model.model."
        if any(k.startswith(tk) for k in sd):
            if getattr(cs, "model", None) is None and hasattr(cs, "build_tower"):
                layers = 1 + max(int(k[len(tk + "transformer.resblocks."):].split(".")[0]) for k in sd
                                 if k.startswith(tk + "transformer.resblocks."))
                cs.build_tower(layers=layers, vocab_size=sd[tk + "token_embedding.weight"].shape[0])
        elif context is not None:
            cs.set_context(context.float())
        elif hasattr(cs, "set_context"):
            cs.require_real_context = True               # forward() raises instead of inventing a context
        m.register_schedule(given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=0.00085, linear_end=0.0120,
                            cosine_s=8e-3)               # 1000-long buffers, as stored
        missing, unexpected = m.load_state_dict(sd, strict=False)
        self._setup_schedule(self.ddpm_steps)
        if vqgan_ckpt is not None:
            self.vq_model.init_from_ckpt(vqgan_ckpt)
        if verbose:
            print(f"[mgld] checkpoint: {len(sd)} entries, {len(missing)} missing, {len(unexpected)} unexpected keys")
            for name, keys in (("missing", missing), ("unexpected", unexpected)):
                if keys:
                    print(f"[mgld]   {name}: " + ", ".join(list(keys)[:6]) + (" ..." if len(keys) > 6 else ""))
        m._graph_key = None
        return missing, unexpected

    def clone_shared(self):
        """Another instance for `SegmentPool` that SHARES this one's parameter and buffer tensors (host memory and build time of one model
        for k segments in flight) but owns its module objects — engines, packed device weights, caches, hipGraphs and streams hang off
        those and stay per instance.  Call it before the first launch of this instance (nothing device-side to copy then)."""
        if self.model._engine is not None:
            raise RuntimeError("clone_shared(): call before the first launch of the source pipeline (its modules already carry an engine)")
        memo = {}
        for mod in (self.model, self.vq_model):
            for t in list(mod.parameters()) + list(mod.buffers()):
                memo[id(t)] = t                 # deepcopy hands these back instead of copying them
        new = copy.copy(self)
        new.model = copy.deepcopy(self.model, memo)
        new.vq_model = copy.deepcopy(self.vq_model, memo)
        for a in ("_shared_gen", "_shared_gen_seed"):
            new.__dict__.pop(a, None)
        return new

    def engine(self):
        from .engine import Engine
        if self.model._engine is None:
            self.model._engine = Engine(chunk_bytes=self.chunk_bytes)
            for sub in (self.model.model.diffusion_model, self.model.structcond_stage_model, self.model.first_stage_model,
                        self.vq_model, self.model.cond_stage_model, self.model.flownet_model):
                sub.set_engine(self.model._engine)
        return self.model._engine

    @torch.no_grad()
    def estimate_flows(self, frames):
        """Flow + occlusion inputs of the guidance from the frames themselves, as the script prepares them
        (oldcanvas_tile.py:392-413): [0,1] quarter-resolution frames -> RAFT_SR both directions (compute_flow) -> resize to
        the latent grid -> forward/backward consistency masks.  frames: [T,3,H,W] in [-1,1].  Returns (flows, masks) in
        the layout run_segment / sample() take."""
        from . import preproc
        from .flowops import forward_backward_consistency_check, resize_flow
        eng = self.engine()
        x = frames.to(eng.device, torch.float32).contiguous()
        H, W = x.shape[-2:]
        lr = preproc.flow_input(x)
        f_fwd, f_bwd = self.model.compute_flow(lr[None])                 # flows[0], flows[1] of the script
        f0 = resize_flow(f_fwd[0], "shape", (H // 8, W // 8))
        f1 = resize_flow(f_bwd[0], "shape", (H // 8, W // 8))
        fo, bo = forward_backward_consistency_check(f1, f0)             # fwd_flow = flows[1], bwd_flow = flows[0] (:405-407)
        return (f0[None], f1[None]), (fo[None, :, None], bo[None, :, None])

    @torch.no_grad()
    def run_segment(self, frames, flows=None, masks=None, guidance_scale=-10.0, noise=None, tile=None, use_graph=True,
                    return_latents=False, shard=None, gather=True, clamp01=True, init_from_vq=False, tile_shard=None):
        """frames: [T,3,H,W] in [-1,1] (the bicubically pre-upsampled LR segment, device or host);
        flows/masks as the reference passes them to sample(); noise: optional dict with 'posterior' [T,4,h,w],
        'x_T' [T,4,h,w], 'steps' [S,T,4,h,w].  Returns HR frames [T,3,H,W] in [0,1] on the device.

        frames may hold k * num_frames frames: k INDEPENDENT segments batched as clips of one pass (round 5; tile=None, unsharded).
        Every network already treats its frame axis as clips of num_frames (the aggregation sampler batches its tiles that way), the
        VAE / AdaIN / posterior arithmetic is per frame, and flows / masks then carry k on their leading axis (one guidance chain
        per clip) — each clip's result is what it produces alone, while every launch sees k x the rows (the 16^2 / 8^2 levels and
        the projections stop being launch-bound: bench.py --clips).

        shard: parallel.FrameShard — the T frames of this segment are split over the ranks (every rank passes the SAME
        full-clip arguments and works on frames [f0, f1)); temporal convolutions exchange one-frame halos, temporal
        attention and the guidance chain all-gather.  With gather=True every rank returns the whole clip."""
        eng = self.engine()
        m, vq = self.model, self.vq_model
        eng.shard = shard
        try:
            return self._run_segment(eng, frames, flows, masks, guidance_scale, noise, tile, use_graph, return_latents,
                                     shard, gather, clamp01, init_from_vq, tile_shard)
        finally:
            eng.shard = None
            eng.tile_shard = None

    def _rank_shared_generator(self):
        """host generator of the noise a SHARDED segment draws itself (no `noise=`): seeded once from the process seed — which every
        rank sets alike — and then ADVANCED from segment to segment, as the reference's global generator is (fresh noise per segment,
        oldcanvas_tile.py:432-436); re-seeded only when the process seed itself changes.  Every rank makes the same calls in the
        same order, so all ranks see the same draws."""
        seed = int(torch.initial_seed()) & 0x7FFFFFFF
        if getattr(self, "_shared_gen_seed", None) != seed:
            self._shared_gen, self._shared_gen_seed = torch.Generator().manual_seed(seed), seed
        return self._shared_gen

    def draw_noise(self, T, h, w):
        """the three draws run_segment makes when no noise is injected — posterior sample (host generator, as
        DiagonalGaussianDistribution.sample does), x_T (device generator, `randn_like(init_latent)`), the per-step noise (device) — in
        that order and from the same generators, as a dict for `run_segment(noise=...)`.  A harness that keeps several segments in
        flight (SegmentPool) draws them HERE, on its own thread, segment by segment: the results then match the one-at-a-time loop."""
        dev = self.engine().device
        shape = (int(T), 4, int(h), int(w))
        post = torch.randn(shape)
        return {"posterior": post, "x_T": torch.randn(shape, device=dev), "steps": torch.randn((self.ddpm_steps,) + shape, device=dev)}

    def _run_segment(self, eng, frames, flows, masks, guidance_scale, noise, tile, use_graph, return_latents, shard, gather,
                     clamp01=True, init_from_vq=False, tile_shard=None):
        m, vq = self.model, self.vq_model
        noise = dict(noise or {})
        fs = None          # frame split of the VAE work around a tile-sharded sampler
        if tile_shard is not None:
            if tile is None or shard is not None:
                raise ValueError("tile_shard belongs to aggregation sampling (tile=...) and excludes frame sharding of the sampler")
            if tile_shard.world > 1 and frames.shape[0] % tile_shard.world == 0:
                from .parallel import FrameShard
                fs = FrameShard(frames.shape[0], tile_shard.rank, tile_shard.world)
            lat_shape = (frames.shape[0], 4, frames.shape[2] // 8, frames.shape[3] // 8)
            g = self._rank_shared_generator()                                              # the same draws on every rank
            for k, shp in (("posterior", lat_shape), ("x_T", lat_shape), ("steps", (self.ddpm_steps,) + lat_shape)):
                if noise.get(k) is None:
                    noise[k] = torch.randn(shp, generator=g)
        if shard is not None:
            if tile is not None:
                raise NotImplementedError("frame sharding and aggregation sampling are separate multi-GPU schemes")
            if frames.shape[0] != shard.T:
                raise ValueError(f"segment has {frames.shape[0]} frames, shard was built for {shard.T}")
            # noise the caller did not inject is drawn for the WHOLE clip from a generator every rank seeds alike, then
            # sliced: the result does not depend on how many ranks share the segment
            lat_shape = (shard.T, 4, frames.shape[2] // 8, frames.shape[3] // 8)
            g = self._rank_shared_generator()
            for k, shp in (("posterior", lat_shape), ("x_T", lat_shape), ("steps", (self.ddpm_steps,) + lat_shape)):
                if noise.get(k) is None:
                    noise[k] = torch.randn(shp, generator=g)
            frames = shard.local(frames)
            for k, dim in (("posterior", 0), ("x_T", 0), ("steps", 1)):
                noise[k] = shard.local(noise[k], dim)
        x = frames.to(eng.device, torch.float32).contiguous()
        T = x.shape[0]
        enc_fea = None
        pn = noise.get("posterior")
        if fs is not None:         # first-stage encode of this rank's frames, latents of the whole clip by all-gather
            post = m.encode_first_stage(fs.local(x).contiguous())
            init_latent = fs.all_gather(m.get_first_stage_encoding(post, fs.local(pn)).contiguous())
            post = None
        else:
            if init_from_vq:
                post, enc_fea = vq.encode(x, hp=True)
            else:
                post = m.encode_first_stage(x)
            init_latent = None
        ctx = m.cond_stage_model([""])
        n0 = noise.get("x_T")
        if fs is None:
            # posterior sample * scale_factor and its q_sample at t = 999 in ONE launch (mgld_init_latent; ddpm.py:3382-3389, 403-406)
            from . import hip
            pn = (pn if pn is not None else torch.randn(post.mean.shape)).to(eng.device, torch.float32).contiguous()   # drawn on the host, as the reference does
            n0 = (torch.randn(pn.shape, device=eng.device) if n0 is None else n0.to(eng.device, torch.float32)).contiguous()
            init_latent, x_T = hip.init_latent(post.parameters.contiguous(), pn, float(m.scale_factor), n0,
                                               self._sa999, self._soma999)
        else:
            n0 = torch.randn_like(init_latent) if n0 is None else n0.to(eng.device)
            t = torch.full((T,), 999, dtype=torch.long, device=eng.device)
            x_T = m.q_sample_respace(init_latent, t, self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod, n0)
        self.last_init_latent = init_latent     # (tests: the struct-cond latent the sampler was handed, next to its output)
        kw = dict(cond=ctx, struct_cond=init_latent, guidance_scale=guidance_scale, flows=flows, masks=masks, batch_size=1,
                  timesteps=self.ddpm_steps, time_replace=self.ddpm_steps, x_T=x_T, noise=noise.get("steps"), use_graph=use_graph)
        if tile is None:
            samples = m.sample(**kw)
        else:
            eng.tile_shard = tile_shard
            try:
                samples = m.sample_canvas(tile_size=tile[0], tile_overlap=tile[1], batch_size_sample=1, **kw)
            finally:
                eng.tile_shard = None
        if fs is not None:         # video VAE on this rank's frames: the 13 temporal convolutions exchange one-frame halos
            eng.shard = fs
            x, z_loc = fs.local(x).contiguous(), fs.local(samples).contiguous()
        else:
            z_loc = samples
        if enc_fea is None:
            _, enc_fea = vq.encode(x)
        x_samples = vq.decode(z_loc * (1.0 / m.scale_factor), enc_fea)
        if self.colorfix_type == "adain":
            x_samples = adaptive_instance_normalization(x_samples, x)
        elif self.colorfix_type == "wavelet":
            x_samples = wavelet_reconstruction(x_samples, x)
        # clamp01=False: colour-fixed frames in [-1,1] as they are (the script's large-image branch averages overlapping
        # patches BEFORE the final clamp, oldcanvas_tile.py:469-471)
        if clamp01:
            from . import hip
            out = hip.to01(x_samples.contiguous())
        else:
            out = x_samples
        if shard is not None and gather and shard.world > 1:
            out, samples = shard.all_gather(out), shard.all_gather(samples)
        if fs is not None:
            out = fs.all_gather(out.contiguous())
        return (out, samples) if return_latents else out


class SegmentPool:
    """Several segments in flight on ONE GPU: `k` pipeline instances, each driven by its own host thread on its own stream.

    A segment's 8 x 64x64-latent launches leave the 256 CUs partly idle again and again — tile counts that do not divide the block
    slots (640 blocks on 512), latency-bound 8x8 / 16x16 levels, norm kernels — and a second or third independent segment fills
    exactly those holes: +19..24 % frames/s at 8 x 512^2 with 2-3 in flight.  Every instance owns its engine,
    arena, hipGraph and sampler stream, and the kernel library keeps the split-K scratch per host thread, so concurrent segments
    share nothing mutable; each result is bit-identical to the one-at-a-time loop (tests/test_nets_gpu.py).  The reference's own
    multi-sequence loop (one process per GPU, sequences dealt round-robin, oldcanvas_tile.py:337-339) is the k = 1 case.

    `make_pipeline()` must return a fresh VSRPipeline (same weights in every instance); jobs are (args, kwargs) of run_segment and
    should inject their noise (`noise=`) so that results do not depend on which worker drew from the global generator first."""

    def __init__(self, make_pipeline, k, first=None, others=None):
        import queue
        import threading
        # instances: `first` (optional, may already be warm), then `others` (prebuilt, e.g. first.clone_shared() taken before its first
        # launch), then fresh ones from make_pipeline()
        self.pipes = ([first] if first is not None else []) + list(others or [])[:max(0, k - (1 if first is not None else 0))]
        self.pipes += [make_pipeline() for _ in range(k - len(self.pipes))]
        self.streams = [torch.cuda.Stream() for _ in self.pipes]
        self.device = torch.cuda.current_device()
        # PERSISTENT workers: one host thread per instance for the pool's lifetime.  The kernel library keeps the split-K scratch per
        # host thread, so a worker registers its 256 MB workspace ONCE (threads started per call re-allocated it inside the timed
        # region, and a graph cached across calls would have replayed against freed scratch).
        self._queues = [queue.Queue() for _ in self.pipes]
        self._threads = [threading.Thread(target=self._worker, args=(i,), daemon=True, name=f"mgld-segment-{i}") for i in range(len(self.pipes))]
        self._sem = threading.Semaphore(0)
        # Entry offset between the workers.  Identical segments entered at the same instant run their launch sequences in lock-step — all in
        # the 64x64-level convolutions together, all in the latency-bound 8x8 level together — and there is nothing left to fill the
        # holes with; ANY offset of 2-26 ms breaks it: 670-675 -> 657 ms per segment at three in flight (profiles/r03_stagger.txt).
        # Round 4 (faster kernels): 20 ms is the best offset on two boxes, -0.5 .. -1.5 % against 4 ms (profiles/r04_stagger.txt).
        self.stagger_ms = float(os.environ.get("MGLD_STAGGER_MS", "20"))
        self.last_latency_ms = {}     # slot -> GPU-side latency of that job in the last _drive call (hipEvent pair on the worker's stream)
        self.closed = False
        self.broken = False           # a worker thread died: no more jobs, close() still joins the others
        for t in self._threads:
            t.start()

    # the worker threads keep the pool (k pipeline instances, a 256 MB split-K scratch per thread) alive: close() it — or use it as a
    # context manager — when the segments are done
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __len__(self):
        return len(self.pipes)

    def _worker(self, i):
        from . import hip
        torch.cuda.set_device(self.device)
        ws_ready = False
        while True:
            item = self._queues[i].get()
            if item is None:
                return
            plan, ready, out, lat, errs = item
            try:
                self.streams[i].wait_event(ready)                # whatever the caller enqueued on ITS stream is complete first
                evs = []
                with torch.cuda.stream(self.streams[i]):
                    if not ws_ready:
                        hip.ensure_workspace()                   # this thread's split-K scratch, for the thread's lifetime
                        ws_ready = True
                    if self.stagger_ms > 0 and i > 0:            # worker i enters the GPU i * stagger_ms after worker 0 (see __init__)
                        time.sleep(1e-3 * self.stagger_ms * i)
                    for slot, call in plan:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        out[slot] = call(self.pipes[i])
                        e1.record()
                        evs.append((slot, e0, e1))
                self.streams[i].synchronize()
                for slot, e0, e1 in evs:
                    lat[slot] = e0.elapsed_time(e1)
            except BaseException as e:   # noqa: BLE001  (re-raised on the calling thread)
                errs.append(e)
            finally:
                self._sem.release()

    def close(self):
        """stop and join the workers.  Also after a worker died (`_drive` marks the pool broken, not closed): the SURVIVING workers
        still get their sentinel and are joined, so their threads, pipeline instances and split-K scratch go away with the pool."""
        if self.closed:
            return
        self.closed = True
        for q in self._queues:
            q.put(None)
        for t in self._threads:
            t.join(timeout=5)
        self._threads = []

    def __del__(self):
        try:
            for q in self._queues:
                q.put(None)
        except Exception:   # noqa: BLE001  (interpreter shutdown)
            pass

    def _drive(self, plan):
        """plan[i] = list of (slot, call) for worker i, call(pipeline instance) -> result; returns {slot: result}.  The GPU-side latency
        of every job (first launch .. last kernel, measured with a hipEvent pair on the worker's stream) lands in last_latency_ms."""
        if self.closed or self.broken:
            raise RuntimeError("SegmentPool is closed" if self.closed else "SegmentPool is broken (a worker thread died): close() it")
        out, lat, errs = {}, {}, []
        ready = torch.cuda.Event()
        ready.record()                     # whatever the caller enqueued for the jobs (inputs, pre-drawn noise) on ITS stream ...
        n = 0
        for i, pl in enumerate(plan):
            if pl:
                self._queues[i].put((pl, ready, out, lat, errs))
                n += 1
        for _ in range(n):
            while not self._sem.acquire(timeout=5.0):      # a worker that died outside its try block would never release: notice it
                if not all(t.is_alive() for t in self._threads):
                    self.broken = True                     # (not `closed`: close() must still shut the surviving workers down)
                    raise RuntimeError("SegmentPool: a worker thread died" + (f": {errs[0]!r}" if errs else ""))
        self.last_latency_ms = lat
        if errs:
            raise errs[0]
        return out

    @staticmethod
    def _seg(job):
        a, kw = job
        return lambda pipe: pipe.run_segment(*a, **kw)

    def run_on(self, i, jobs):
        """jobs (args, kwargs of run_segment) on instance i only, in order (warm-up: fills that instance's caches)"""
        res = self._drive([[(j, self._seg(job)) for j, job in enumerate(jobs)] if w == i else [] for w in range(len(self.pipes))])
        return [res[j] for j in range(len(jobs))]

    def map_on(self, i, fn, items):
        """results[j] = fn(pipeline instance i, items[j]) on instance i only, in order"""
        res = self._drive([[(j, (lambda pipe, it=it: fn(pipe, it))) for j, it in enumerate(items)] if w == i else [] for w in range(len(self.pipes))])
        return [res[j] for j in range(len(items))]

    def map(self, fn, items):
        """results[j] = fn(pipeline instance, items[j]); item j on instance j % k, every instance on its own thread + stream"""
        k = len(self.pipes)
        res = self._drive([[(j, (lambda pipe, it=it: fn(pipe, it))) for j, it in enumerate(items) if j % k == i] for i in range(k)])
        return [res[j] for j in range(len(items))]

    def run(self, jobs):
        """jobs[j] = (args, kwargs) of run_segment -> results[j]; static round-robin (equal segments need no work queue)"""
        return self.map(lambda pipe, job: pipe.run_segment(*job[0], **job[1]), jobs)
