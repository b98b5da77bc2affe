"""The fixed-size entry scripts of the reference, scripts/vsr_val_ddpm_text_T_vqganfin_old.py and ..._w_latent.py, as one
implementation (`main(argv, w_latent)`): same option surface and defaults, same per-segment call sequence, model work on libmgld_hip.

What these scripts do differently from the tiled README script (SURVEY.md Appendix A "Input geometry rules"):
  * frames are Lanczos-resized on the host to multiples of 32 (PIL, part of the image decode: old.py:107-116), then torchvision
    `Resize(input_size)` + `CenterCrop(input_size)` — here one device kernel (hip.resize_center_crop) — and clamped (:253-256, 314-316);
  * trailing frames that do not fill a segment are dropped (:307-312), no repeat-last padding;
  * the init latent comes from the VIDEO VAE's own encoder (`vq_model.encode`, :328-329), whose features also feed the decoder;
  * optical flow is estimated on the full-resolution [0,1] frames and resized by ratio 1/8 (:344-349);
  * occlusion masks: old.py takes (fwd_flow, bwd_flow) = (flows[0], flows[1]) (:354), w_latent.py (flows[1], flows[0]) (:360, the
    tiled script's convention); guidance scale -10 (old.py:365) vs -1 (w_latent.py:371);
  * plain `model.sample` (no aggregation sampling); w_latent.py also writes one `<frame>.npy` [4,h,w] latent per frame (:389-397).
Noise is drawn on the device generator (values differ from a CUDA run of the reference, as between any two GPU models); the parity
tests inject the noise.
"""
import argparse
import os
import sys

import numpy as np
import torch

from .pipeline import VSRPipeline, model_configs

IMAGE_EXTS = (".png", ".jpg", ".jpeg", ".bmp", ".tif", ".tiff", ".webp")
REF_CONFIG = "configs/stableSRNew/v2-finetune_text_T_512.yaml"
REF_CKPT = "models/ldm/stable-diffusion-v1/model.ckpt"
REF_VQGAN_CKPT = "models/ldm/stable-diffusion-v1/epoch=000011.ckpt"

NOISE_HOOK = None     # tests: NOISE_HOOK(T, h, w, steps) -> {"posterior", "x_T", "steps"}
CAPTURE = None        # tests: list receiving {"flows", "masks", "x0"} per segment
FLOW_HOOK = None      # tests: FLOW_HOOK(flows, masks) -> (flows, masks) the sampler gets instead (CAPTURE keeps this build's as own_*)


def load_img(path):
    """old.py:107-116: RGB, both sides cut down to multiples of 32 by a LANCZOS resize, [-1,1] NCHW (host)."""
    from PIL import Image
    image = Image.open(path).convert("RGB")
    w, h = image.size
    w, h = w - w % 32, h - h % 32
    image = image.resize((w, h), resample=Image.LANCZOS)
    a = np.asarray(image, dtype=np.float32) / 255.0
    return 2.0 * torch.from_numpy(a).permute(2, 0, 1)[None] - 1.0


def parse(argv, w_latent):
    p = argparse.ArgumentParser()
    p.add_argument("--seqs-path", type=str, nargs="?", default="inputs/user_upload")
    p.add_argument("--outdir", type=str, nargs="?", default="outputs/user_upload")
    if w_latent:
        p.add_argument("--latent-dir", type=str, nargs="?", default="latent/user_upload")
    p.add_argument("--ddpm_steps", type=int, default=200)
    p.add_argument("--C", type=int, default=4)
    p.add_argument("--f", type=int, default=8)
    p.add_argument("--n_frames", type=int, default=5)
    p.add_argument("--n_samples", type=int, default=1)
    p.add_argument("--config", type=str, default=REF_CONFIG)
    p.add_argument("--vqgan_config", type=str, default=None, help="(extension) video-VAE YAML; default = shipped hyper-parameters")
    p.add_argument("--ckpt", type=str, default=REF_CKPT)
    p.add_argument("--vqgan_ckpt", type=str, default=REF_VQGAN_CKPT)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--precision", type=str, default="autocast", choices=["full", "autocast"])
    p.add_argument("--select_idx", type=int, default=0)
    p.add_argument("--n_gpus", type=int, default=1)
    p.add_argument("--input_size", type=int, default=512)
    p.add_argument("--dec_w", type=float, default=0.5)
    p.add_argument("--colorfix_type", type=str, default="nofix")
    p.add_argument("--inflight", type=int, default=1,
                   help="(extension) segments kept in flight on the GPU (pipeline.SegmentPool: one model instance + host thread + stream "
                        "each; 2-3 give ~+20 %% frames/s).  The noise of every segment is drawn on the main thread in segment order, "
                        "so the output does not depend on this option")
    opt = p.parse_args(argv)
    if opt.C != 4 or opt.f != 8:
        p.error("--C / --f: the SD-2.1 latent space of this path is 4 channels at 1/8 resolution")
    if opt.input_size % 32:
        p.error("--input_size must be a multiple of 32")
    for name, ref in (("config", REF_CONFIG), ("ckpt", REF_CKPT), ("vqgan_ckpt", REF_VQGAN_CKPT)):
        v = getattr(opt, name)
        if v is not None and not os.path.exists(v):
            if v == ref:
                print(f"[mgld] --{name}: default file {ref} not found -> built-in {'hyper-parameters' if name == 'config' else 'synthetic weights'}")
                setattr(opt, name, None)
            else:
                p.error(f"--{name}: {v} does not exist")
    return opt


def _yaml_model(path):
    import yaml
    with open(path) as fh:
        return yaml.safe_load(fh)["model"]


def main(argv=None, w_latent=False):
    from . import hip
    from .flowops import forward_backward_consistency_check, resize_flow
    opt = parse(argv, w_latent)
    torch.manual_seed(opt.seed)                                        # seed_everything(opt.seed), once (:250)
    cfgs = model_configs(opt.n_frames)
    if opt.config:
        d = _yaml_model(opt.config)
        d["params"].pop("ckpt_path", None)
        d["params"]["first_stage_config"]["params"].pop("ckpt_path", None)
        cfgs = (d, cfgs[1])
    if opt.vqgan_config:
        v = _yaml_model(opt.vqgan_config)
        v["params"].pop("ckpt_path", None)
        v["params"]["lossconfig"] = {"target": "torch.nn.Identity"}
        cfgs = (cfgs[0], v)
    def make_pipe():
        pp = VSRPipeline(num_frames=opt.n_frames, ddpm_steps=opt.ddpm_steps, dec_w=opt.dec_w, colorfix_type=opt.colorfix_type,
                         synthetic_weights=opt.ckpt is None, configs=cfgs)
        pp.load_weights(opt.ckpt, opt.vqgan_ckpt)
        return pp

    pipe = make_pipe()
    os.makedirs(opt.outdir, exist_ok=True)
    from . import preproc
    from .preproc import FrameWriter
    writer = FrameWriter()
    gscale = -1.0 if w_latent else -10.0
    h8 = opt.input_size // 8

    def segment_job(pp, item):
        """one segment on pipeline instance `pp` (the calling thread's current stream): decode -> resize/crop -> flows -> sample -> payload"""
        seq, seg_names, nz = item
        dev = pp.engine().device
        frames = []
        for f in seg_names:
            img = load_img(os.path.join(opt.seqs_path, seq, f)).to(dev)
            frames.append(torch.clamp(hip.resize_center_crop(img, opt.input_size), -1.0, 1.0))
        x = torch.cat(frames, 0)                                                       # [T,3,S,S] in [-1,1]
        x01 = torch.clamp((x + 1.0) / 2.0, min=0.0, max=1.0)
        f0, f1 = pp.model.compute_flow(x01[None])                                      # full-resolution flows (:342-344)
        f0, f1 = resize_flow(f0[0], "ratio", (0.125, 0.125)), resize_flow(f1[0], "ratio", (0.125, 0.125))
        fwd, bwd = (f1, f0) if w_latent else (f0, f1)                                  # (:354) vs w_latent (:360)
        fo, bo = forward_backward_consistency_check(fwd, bwd)
        flows, masks = (f0[None], f1[None]), (fo[None, :, None], bo[None, :, None])
        own = (flows, masks)
        if FLOW_HOOK is not None:
            flows, masks = FLOW_HOOK(flows, masks)
        out, lat = pp.run_segment(x, flows=flows, masks=masks, guidance_scale=gscale, noise=nz, return_latents=True, init_from_vq=True)
        cap = {"flows": flows, "masks": masks, "x0": lat, "frames": x, "own_flows": own[0], "own_masks": own[1]} if CAPTURE is not None else None
        arrs = preproc.to_png_payload(out, opt.input_size, opt.input_size)
        return arrs, (lat.cpu().numpy() if w_latent else None), cap

    def emit(item, res):
        seq, seg_names, _ = item
        arrs, lat_np, cap = res
        if cap is not None:
            CAPTURE.append(cap)
        for k, f in enumerate(seg_names):
            base = os.path.splitext(os.path.basename(f))[0]
            writer.png(os.path.join(opt.outdir, seq, base + ".png"), arrs[k])           # encoded off this thread
            if w_latent:
                writer.npy(os.path.join(opt.latent_dir, seq, base + ".npy"), lat_np[k])

    pool = None
    if opt.inflight > 1:
        from .pipeline import SegmentPool
        rng_cpu, rng_dev = torch.get_rng_state(), torch.cuda.get_rng_state()
        # the extra instances share the first one's host weights when it has not launched anything yet (else: built from scratch)
        shared = [pipe.clone_shared() for _ in range(opt.inflight - 1)] if pipe.model._engine is None else None
        pool = SegmentPool(make_pipe, opt.inflight, first=pipe, others=shared)
        torch.set_rng_state(rng_cpu)          # building the extra instances draws from the global generators (module initialisers):
        torch.cuda.set_rng_state(rng_dev)     # put them back, so that the segments' noise does not depend on --inflight
    pending = []

    def flush():
        if not pending:
            return
        results = pool.map(segment_job, pending) if pool is not None else [segment_job(pipe, it) for it in pending]
        for it, res in zip(pending, results):
            emit(it, res)
        pending.clear()

    for seq_idx, seq in enumerate(sorted(os.listdir(opt.seqs_path))):
        if seq_idx % opt.n_gpus != opt.select_idx:                     # process-level sharding (:302-303)
            continue
        names = sorted(f for f in os.listdir(os.path.join(opt.seqs_path, seq)) if f.lower().endswith(IMAGE_EXTS))
        if not names:
            print(f"[mgld] {os.path.join(opt.seqs_path, seq)}: no image files - skipped")
            continue
        n_seg = len(names) // opt.n_frames                             # a trailing partial segment is dropped (:307-312)
        os.makedirs(os.path.join(opt.outdir, seq), exist_ok=True)
        if w_latent:
            os.makedirs(os.path.join(opt.latent_dir, seq), exist_ok=True)
        for n in range(n_seg):
            seg_names = names[n * opt.n_frames:(n + 1) * opt.n_frames]
            # the segment's noise, drawn HERE in segment order (the draws run_segment would make itself, same generators): what a
            # segment gets does not depend on how many are in flight
            nz = NOISE_HOOK(len(seg_names), h8, h8, opt.ddpm_steps) if NOISE_HOOK is not None else pipe.draw_noise(len(seg_names), h8, h8)
            pending.append((seq, seg_names, nz))
            if len(pending) >= max(1, opt.inflight) * (4 if pool is not None else 1):
                flush()
    flush()
    writer.close()
    if pool is not None:
        pool.close()
    return 0


if __name__ == "__main__":
    # `python -m mgld_vsr_amd.cli_simple [--w-latent] ...`: the flag selects the _w_latent script's behaviour and is not an argparse option
    sys.exit(main([a for a in sys.argv[1:] if a != "--w-latent"], w_latent="--w-latent" in sys.argv[1:]))
