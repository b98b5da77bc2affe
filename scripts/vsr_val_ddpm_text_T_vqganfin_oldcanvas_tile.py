#!/usr/bin/env python
"""MI355X counterpart of the reference's README inference entry
(scripts/vsr_val_ddpm_text_T_vqganfin_oldcanvas_tile.py:133-556): same argument surface, same segmenting / padding /
resizing rules, same per-segment call sequence — with the model work done by libmgld_hip.  Only the image file decode /
encode (PIL) runs on the host; the bicubic pre-upsampling, reflect padding, quarter-resolution flow input, RAFT_SR flow
estimation, occlusion masks, patch splitting / blending and the uint8 conversion run on the device (mgld_vsr_amd/preproc.py,
raft.py, flowops.py).

Differences that are deliberate (SURVEY.md §3.1 "known defects"):
  * `--flows-path` (extension): precomputed flows (`<seq>/<segment>_flows.npy`, array [2, T-1, 2, h/4, w/4] =
    (flows_forward_prop, flows_backward_prop) as `compute_flow` returns them) instead of RAFT_SR;
  * the reference's non-tiled branch reads `flow_f/flow_b/fwd_occ/bwd_occ` before assignment (:492-495); here the
    evident intent (`flows[0], flows[1], fwd_occs, bwd_occs`) is used;
  * the VAE config path that does not exist in the reference (:303) is replaced by `--vqgan_config`;
  * seeding follows the reference: `seed_everything(opt.seed)` once at start (:280) and again per pixel patch in the
    large-frame branch (:428) — on the device generator here, so the noise VALUES differ from a CUDA run of the reference
    (they would between two GPU models as well); parity tests inject the noise instead.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from mgld_vsr_amd.pipeline import VSRPipeline, model_configs  # noqa: E402


def load_yaml_model_section(path):
    import yaml
    with open(path) as fh:
        cfg = yaml.safe_load(fh)
    return cfg["model"]


def read_image(path):
    from PIL import Image
    im = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32) / 255.0
    t = torch.from_numpy(im).permute(2, 0, 1)[None]
    return (t - 0.5) / 0.5


# option names and defaults as in the reference script (:134-:262)
IMAGE_EXTS = (".png", ".jpg", ".jpeg", ".bmp", ".tif", ".tiff", ".webp")
REF_CONFIG = "configs/stable-diffusion/v1-inference.yaml"
REF_CKPT = "checkpoints/stablevsr_025.ckpt"
REF_VQGAN_CKPT = "checkpoints/vqgan_cfw_00011.ckpt"


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--seqs-path", type=str, nargs="?", default="inputs/user_upload",
                   help="dir with one sub-dir of LR PNG frames per sequence")
    p.add_argument("--flows-path", type=str, default=None, help="(extension) precomputed flows instead of RAFT_SR")
    p.add_argument("--outdir", type=str, nargs="?", default="outputs/user_upload")
    p.add_argument("--device", type=str, default="cuda", help="accepted for compatibility; this path runs on the HIP device only")
    p.add_argument("--ddpm_steps", type=int, default=1000)
    p.add_argument("--n_iter", type=int, default=1, help="accepted for compatibility (unused by the reference script as well)")
    p.add_argument("--C", type=int, default=4, help="latent channels (4)")
    p.add_argument("--f", type=int, default=8, help="downsampling factor (8)")
    p.add_argument("--n_frames", type=int, default=5)
    p.add_argument("--n_samples", type=int, default=1)
    p.add_argument("--config", type=str, default=REF_CONFIG,
                   help="diffusion YAML (model: section); the shipped hyper-parameters when the default file is absent")
    p.add_argument("--vqgan_config", type=str, default=None, help="(extension) video-VAE YAML; default = shipped hyper-parameters")
    p.add_argument("--ckpt", type=str, default=REF_CKPT, help="synthetic weights (with a warning) when the default file is absent")
    p.add_argument("--vqgan_ckpt", type=str, default=REF_VQGAN_CKPT)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--inflight", type=int, default=1,
                   help="large frames: pixel patches of a segment kept in flight on this GPU (pipeline.SegmentPool; 1 = the reference's "
                        "loop).  The reference re-seeds before every patch, so each patch's noise is a function of the seed and its shape "
                        "alone: drawn on the main thread, patch by patch, the output does not depend on this option.")
    p.add_argument("--precision", type=str, default="autocast", choices=["full", "autocast"])
    p.add_argument("--select_idx", type=int, default=0)
    p.add_argument("--n_gpus", type=int, default=1)
    p.add_argument("--dec_w", type=float, default=0.5)
    p.add_argument("--tile_overlap", type=int, default=32)
    p.add_argument("--upscale", type=float, default=4.0)
    p.add_argument("--colorfix_type", type=str, default="nofix", choices=["adain", "wavelet", "nofix"])
    p.add_argument("--vqgantile_stride", type=int, default=750)
    p.add_argument("--vqgantile_size", type=int, default=960)
    p.add_argument("--guidance_scale", type=float, default=-10.0, help="(extension) the reference hard-codes -10")
    p.add_argument("--latent-dir", type=str, default=None,
                   help="also dump the sampled latents, one <frame>.npy [4,h,w] per frame (what vsr_val_ddpm_text_T_vqganfin_w_latent.py"
                        ":389-397 writes for the stage-2 VAE-decoder training); frames that fit one patch only")
    opt = p.parse_args(argv)
    if opt.device != "cuda":
        p.error("--device: only 'cuda' (the HIP device) is supported; there is no CPU path")
    if opt.C != 4 or opt.f != 8:
        p.error("--C / --f: the SD-2.1 latent space of this path is 4 channels at 1/8 resolution")
    # the reference's default files are not shipped: fall back to the built-in hyper-parameters / synthetic weights when a
    # DEFAULT path is absent; an explicitly given path must exist
    for name, ref in (("config", REF_CONFIG), ("ckpt", REF_CKPT), ("vqgan_ckpt", REF_VQGAN_CKPT)):
        v = getattr(opt, name)
        if v is not None and not os.path.exists(v):
            if v == ref:
                print(f"[mgld] --{name}: default file {ref} not found -> built-in {'hyper-parameters' if name == 'config' else 'synthetic weights'}")
                setattr(opt, name, None)
            else:
                p.error(f"--{name}: {v} does not exist")
    return opt


# test hooks (tests/test_cli_gpu.py): NOISE_HOOK(T, h, w, steps) -> the noise dict run_segment takes (the parity test replays the
# draws of a captured reference run); CAPTURE: a list that receives, per sampled patch, its flows / masks / sampled latents;
# FLOW_HOOK(patch index, flows, masks) -> (flows, masks) the sampler gets instead (the production-schedule parity test hands one patch
# the reference run's flows, to separate the sampler's deviation from what RAFT's fp16 flows add)
NOISE_HOOK = None
CAPTURE = None
FLOW_HOOK = None


def main(argv=None):
    opt = parse(argv)
    torch.manual_seed(opt.seed)
    cfgs = model_configs(opt.n_frames)
    if opt.config:
        cfgs = (load_yaml_model_section(opt.config), cfgs[1])
        cfgs[0]["params"].pop("ckpt_path", None)
        cfgs[0]["params"]["first_stage_config"]["params"].pop("ckpt_path", None)
    if opt.vqgan_config:
        v = load_yaml_model_section(opt.vqgan_config)
        v["params"].pop("ckpt_path", None)
        v["params"]["lossconfig"] = {"target": "torch.nn.Identity"}
        cfgs = (cfgs[0], v)
    pipe = VSRPipeline(num_frames=opt.n_frames, ddpm_steps=opt.ddpm_steps, dec_w=opt.dec_w,
                       colorfix_type=opt.colorfix_type, synthetic_weights=opt.ckpt is None, configs=cfgs)
    pipe.load_weights(opt.ckpt, opt.vqgan_ckpt)       # 1000-step buffers first, respacing after; text tower from the checkpoint; refuses
                                                      # a real --ckpt without a video-VAE checkpoint (the VAE would run on zeros)
    pool = None
    if opt.inflight > 1:                         # extra instances share the first one's host weights (taken before its first launch)
        from mgld_vsr_amd.pipeline import SegmentPool
        pool = SegmentPool(None, opt.inflight, first=pipe, others=[pipe.clone_shared() for _ in range(opt.inflight - 1)])
    os.makedirs(opt.outdir, exist_ok=True)
    from mgld_vsr_amd.preproc import FrameWriter
    writer = FrameWriter()                          # PNG / .npy encoding overlaps the next segment's sampling

    seq_names = sorted(os.listdir(opt.seqs_path))
    for seq_idx, seq in enumerate(seq_names):
        if seq_idx % opt.n_gpus != opt.select_idx:   # the reference's process-level sharding (:337-339)
            continue
        # every image file of the sequence directory, sorted by name (the reference takes sorted(os.listdir), :343)
        paths = sorted(os.path.join(opt.seqs_path, seq, f) for f in os.listdir(os.path.join(opt.seqs_path, seq))
                       if f.lower().endswith(IMAGE_EXTS))
        if not paths:
            print(f"[mgld] {os.path.join(opt.seqs_path, seq)}: no image files ({', '.join(IMAGE_EXTS)}) - skipped")
            continue
        while len(paths) % opt.n_frames:             # repeat-last padding (:345-346)
            paths.append(paths[-1])
        from mgld_vsr_amd import preproc
        lr = [read_image(pth) for pth in paths]      # host: PNG decode only; everything else runs on the device
        os.makedirs(os.path.join(opt.outdir, seq), exist_ok=True)
        for s0 in range(0, len(lr), opt.n_frames):
            up = preproc.upsample_lr(torch.cat(lr[s0:s0 + opt.n_frames], 0), opt.upscale)   # (:349-357), clamped (:376)
            seg, ori_h, ori_w = preproc.pad_to_32(up)                                         # reflect pad (:381-390)
            flows = masks = None
            if opt.flows_path:
                fpath = os.path.join(opt.flows_path, seq, f"{s0 // opt.n_frames:04d}_flows.npy")
                fl = torch.from_numpy(np.load(fpath)).float()
                from basicsr.archs.arch_util import resize_flow
                from scripts.util_flow import forward_backward_consistency_check
                h8, w8 = seg.shape[-2] // 8, seg.shape[-1] // 8
                f_fwd = resize_flow(fl[0], "shape", [h8, w8])
                f_bwd = resize_flow(fl[1], "shape", [h8, w8])
                # fwd_flow = flows[1] (true forward flow), bwd_flow = flows[0]  (oldcanvas_tile.py:405-409)
                fo, bo = forward_backward_consistency_check(f_bwd, f_fwd)
                flows, masks = (f_fwd[None], f_bwd[None]), (fo[None, :, None], bo[None, :, None])
            else:                                      # as the reference: estimate them with RAFT_SR (:392-413)
                flows, masks = pipe.estimate_flows(seg)
            latents = []

            npatch = [0]

            def one(frames, fl, mk, reseed=False):
                if FLOW_HOOK is not None:
                    own = (fl, mk)
                    fl, mk = FLOW_HOOK(npatch[0], fl, mk)
                npatch[0] += 1
                if reseed:
                    torch.manual_seed(opt.seed)                  # seed_everything(opt.seed) per pixel patch (:428)
                h8_, w8_ = frames.shape[-2] // 8, frames.shape[-1] // 8
                tl = None if (h8_ <= 64 and w8_ <= 64) else (64, opt.tile_overlap)   # one 64x64 tile == plain sampling
                nz = NOISE_HOOK(frames.shape[0], h8_, w8_, opt.ddpm_steps) if NOISE_HOOK is not None else None
                out_, lat_ = pipe.run_segment(frames, flows=fl, masks=mk, guidance_scale=opt.guidance_scale, tile=tl, clamp01=False,
                                              return_latents=True, noise=nz)
                latents.append(lat_)
                if CAPTURE is not None:
                    CAPTURE.append({"flows": fl, "masks": mk, "x0": lat_, "lat": getattr(pipe, "last_init_latent", None)})
                    if FLOW_HOOK is not None:
                        CAPTURE[-1]["own_flows"], CAPTURE[-1]["own_masks"] = own
                return out_

            if seg.shape[-2] > opt.vqgantile_size or seg.shape[-1] > opt.vqgantile_size:
                # large frames: overlapping pixel patches through the WHOLE path, uniform-count blending (:418-471)
                from scripts.util_image import ImageSpliterTh
                ps, st = opt.vqgantile_size, opt.vqgantile_stride
                im_sp = ImageSpliterTh(seg, ps, st, sf=1)
                # flows [1,T-1,2,h,w] / masks [1,T-1,1,h,w] -> 4-D [T-1,c,h,w] for the latent-resolution spliters
                aux = [ImageSpliterTh(t[0], ps // 8, st // 8, sf=1) for t in (flows[0], flows[1], masks[0], masks[1])]
                if pool is None:
                    for (pch, idx), (ff_, _), (fb_, _), (fo_, _), (bo_, _) in zip(im_sp, *aux):
                        im_sp.update(one(pch, (ff_[None], fb_[None]), (fo_[None], bo_[None]), reseed=True), idx)
                else:
                    # patches in flight: every patch is re-seeded (:428), so its draws depend on (seed, shape) only — made HERE, in the
                    # order run_segment makes them (VSRPipeline.draw_noise), and injected; the patches then run on the pool's instances
                    jobs = []
                    for (pch, idx), (ff_, _), (fb_, _), (fo_, _), (bo_, _) in zip(im_sp, *aux):
                        torch.manual_seed(opt.seed)
                        h8_, w8_ = pch.shape[-2] // 8, pch.shape[-1] // 8
                        nz = NOISE_HOOK(pch.shape[0], h8_, w8_, opt.ddpm_steps) if NOISE_HOOK is not None else pipe.draw_noise(pch.shape[0], h8_, w8_)
                        fl_, mk_ = (ff_[None], fb_[None]), (fo_[None], bo_[None])
                        if FLOW_HOOK is not None:
                            fl_, mk_ = FLOW_HOOK(len(jobs), fl_, mk_)
                        jobs.append((pch, fl_, mk_, nz, idx))

                    def patch(pipe_i, job):
                        pch, fl, mk, nz, _ = job
                        h8_, w8_ = pch.shape[-2] // 8, pch.shape[-1] // 8
                        tl = None if (h8_ <= 64 and w8_ <= 64) else (64, opt.tile_overlap)
                        return pipe_i.run_segment(pch, flows=fl, masks=mk, guidance_scale=opt.guidance_scale, tile=tl, clamp01=False,
                                                  return_latents=True, noise=nz)
                    for (out_, lat_), job in zip(pool.map(patch, jobs), jobs):
                        latents.append(lat_)
                        if CAPTURE is not None:
                            CAPTURE.append({"flows": job[1], "masks": job[2], "x0": lat_})
                        im_sp.update(out_, job[4])
                x_samples = im_sp.gather()
            else:
                x_samples = one(seg, flows, masks)
            out = torch.clamp((x_samples + 1.0) / 2.0, min=0.0, max=1.0)
            size_min = min(lr[s0].shape[-2:])
            up_scale = max(512.0 / size_min, opt.upscale)
            if up_scale > opt.upscale:                # small inputs were upsampled further: back to the requested scale (:523-530)
                from mgld_vsr_amd import hip
                out = hip.resize_bicubic(out, (int(seg.shape[-2] * opt.upscale / up_scale), int(seg.shape[-1] * opt.upscale / up_scale)),
                                         clamp=(0.0, 1.0))
                ori_h, ori_w = min(ori_h, out.shape[-2]), min(ori_w, out.shape[-1])
            arrs = preproc.to_png_payload(out, ori_h, ori_w)                                  # crop + uint8 (:532-543)
            for k in range(arrs.shape[0]):
                base = os.path.splitext(os.path.basename(paths[s0 + k]))[0]
                writer.png(os.path.join(opt.outdir, seq, base + ".png"), arrs[k])                 # '{}.png'.format(basename) (:541)
            if opt.latent_dir and len(latents) == 1:             # w_latent.py:389-397
                os.makedirs(os.path.join(opt.latent_dir, seq), exist_ok=True)
                lat = latents[0].cpu().numpy()
                for k in range(lat.shape[0]):
                    base = os.path.splitext(os.path.basename(paths[s0 + k]))[0]
                    writer.npy(os.path.join(opt.latent_dir, seq, base + ".npy"), lat[k])
    writer.close()
    if pool is not None:
        pool.close()

if __name__ == "__main__":
    main()
