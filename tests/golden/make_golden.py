"""Generate the golden fixtures under tests/golden/*.npz by running the REFERENCE's own Python
(/root/reference, read-only) in the build container.  Fixtures hold inputs + expected outputs (+ the reference's
parameter names/shapes); weights are never stored — both sides regenerate them from mgld_vsr_amd.synth.

    python tests/golden/make_golden.py            # (re)writes every fixture

Never runs on the GPU box (the reference does not travel); the committed .npz files do.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_import  # noqa: E402
from configs import STRUCT_SMALL, T, UNET_SMALL, VAE_DD_SMALL  # noqa: E402
from mgld_vsr_amd import synth  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(8)


OUT = os.environ.get("MGLD_GOLDEN_OUT", HERE)     # the regenerate-and-compare test writes into a temp dir


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def names_shapes(module):
    return json.dumps([[k, list(v.shape)] for k, v in module.state_dict().items() if v.is_floating_point()])


# ------------------------------------------------------------------------------------------------------------------
def gen_flow():
    au = ref_import.ref("basicsr.archs.arch_util")
    uf = ref_import.ref("scripts.util_flow")
    n, c, h, w = 3, 4, 20, 24
    x = synth.synth_tensor("flow/x", (n, c, h, w))
    flow = synth.smooth_flow("flow/f", n, h, w, amp=3.0)
    flow[0, :, :3] += 40.0  # far out of range
    flow[1, :, :, -2:] += 1.7  # straddles the right border
    with torch.enable_grad():
        xg = x.clone().requires_grad_(True)
        out = au.flow_warp(xg, flow.permute(0, 2, 3, 1))
        up = synth.synth_tensor("flow/up", (n, c, h, w))
        grad = torch.autograd.grad((out * up).sum(), xg)[0]
    out2 = uf.flow_warp(x, flow)
    fwd, bwd = synth.smooth_flow("flow/fw", n, h, w, 2.0), synth.smooth_flow("flow/bw", n, h, w, 2.0)
    bwd = -fwd + 0.3 * bwd
    focc, bocc = uf.forward_backward_consistency_check(fwd, bwd, alpha=0.01, beta=0.5)
    rs = au.resize_flow(fwd, "shape", [h // 2, w // 2])
    rs2 = au.resize_flow(fwd, "ratio", [0.5, 0.5])
    save("g_flow", x=x, flow=flow, warp=out.detach(), up=up, grad=grad, warp_n2hw=out2, fwd=fwd, bwd=bwd, focc=focc, bocc=bocc,
         resized=rs, resized_ratio=rs2)


def gen_guidance():
    ddpm = ref_import.ref("ldm.models.diffusion.ddpm")
    uf = ref_import.ref("scripts.util_flow")
    out = {}
    for Tn, h, w in [(3, 12, 16), (5, 16, 16)]:
        z = synth.synth_tensor(f"guid/z{Tn}", (Tn, 4, h, w), 0.8)
        ff = synth.smooth_flow(f"guid/ff{Tn}", Tn - 1, h, w)
        fb = synth.smooth_flow(f"guid/fb{Tn}", Tn - 1, h, w)
        focc, bocc = uf.forward_backward_consistency_check(fb, ff)
        focc[:, :2] = 1.0
        self_ = types.SimpleNamespace(num_frames=Tn)
        with torch.enable_grad():
            zg = z.clone().requires_grad_(True)
            loss = ddpm.LatentDiffusionVSRTextWT.compute_temporal_condition_v4(
                self_, (ff[None], fb[None]), zg, (focc[None, :, None], bocc[None, :, None]))
            grad = torch.autograd.grad(loss, zg)[0]
        out.update({f"z{Tn}": z, f"ff{Tn}": ff, f"fb{Tn}": fb, f"focc{Tn}": focc, f"bocc{Tn}": bocc,
                    f"loss{Tn}": loss.detach().reshape(1), f"grad{Tn}": grad})
    save("g_guidance", **out)


# ---- reduced full model ---------------------------------------------------------------------------------------------
class _StubCond(torch.nn.Module):
    def __init__(self, ctx_dim=64, **kw):
        super().__init__()
        self.ctx_dim = ctx_dim
        self.device = "cpu"

    def forward(self, text):
        return synth.synth_tensor("ctx", (1, 77, self.ctx_dim)).repeat(len(text), 1, 1)

    def encode(self, text):
        return self(text)


class _StubFlow(torch.nn.Module):
    def __init__(self, **kw):
        super().__init__()


def build_ref_model(unet_cfg=None, struct_cfg=None, vae_dd=None, num_frames=T, flownet_config=None):
    unet_cfg, struct_cfg = dict(unet_cfg or UNET_SMALL), dict(struct_cfg or STRUCT_SMALL)
    stubs = types.ModuleType("golden_stubs")
    stubs.StubCond, stubs.StubFlow = _StubCond, _StubFlow
    sys.modules["golden_stubs"] = stubs
    ddpm = ref_import.ref("ldm.models.diffusion.ddpm")
    fs_dd = dict(vae_dd or VAE_DD_SMALL)
    fs_dd.pop("num_frames")
    model = ddpm.LatentDiffusionVSRTextWT(
        first_stage_config={"target": "ldm.models.autoencoder.AutoencoderKL",
                            "params": {"embed_dim": 4, "ddconfig": fs_dd, "lossconfig": {"target": "torch.nn.Identity"}}},
        cond_stage_config={"target": "golden_stubs.StubCond", "params": {"ctx_dim": unet_cfg["context_dim"]}},
        structcond_stage_config={"target": "ldm.modules.diffusionmodules.openaimodel.InflatedEncoderUNetModelWT",
                                 "params": struct_cfg},
        flownet_config=flownet_config or {"target": "golden_stubs.StubFlow", "params": {}},
        num_frames=num_frames, linear_start=0.00085, linear_end=0.0120, num_timesteps_cond=1, log_every_t=200, timesteps=1000,
        first_stage_key="image", cond_stage_key="caption", image_size=128, channels=4, cond_stage_trainable=False,
        conditioning_key="crossattn", scale_factor=0.18215, use_ema=False, time_replace=1000, use_usm=True,
        unet_config={"target": "ldm.modules.diffusionmodules.openaimodel.InflatedUNetModelDualcondV2",
                     "params": unet_cfg})
    model.eval()
    synth.fill_module_(model.model.diffusion_model, "unet")
    synth.fill_module_(model.structcond_stage_model, "structcond")
    synth.fill_module_(model.first_stage_model, "first_stage")
    model.configs = types.SimpleNamespace(model=types.SimpleNamespace(params=types.SimpleNamespace(channels=4)))
    return model, ddpm


def respace(model, steps):
    """scripts/vsr_val_ddpm_text_T_vqganfin_oldcanvas_tile.py:308-329, verbatim procedure."""
    import copy
    ddpm = ref_import.ref("ldm.models.diffusion.ddpm")
    model.register_schedule(given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=0.00085,
                            linear_end=0.0120, cosine_s=8e-3)
    model.num_timesteps = 1000
    sac = copy.deepcopy(model.sqrt_alphas_cumprod)
    somac = copy.deepcopy(model.sqrt_one_minus_alphas_cumprod)
    use_timesteps = set(ddpm.space_timesteps(1000, [steps]))
    last = 1.0
    new_betas = []
    for i, ac in enumerate(model.alphas_cumprod):
        if i in use_timesteps:
            new_betas.append(1 - ac / last)
            last = ac
    new_betas = [b.data.cpu().numpy() for b in new_betas]
    model.register_schedule(given_betas=np.array(new_betas), timesteps=len(new_betas))
    model.num_timesteps = 1000
    model.ori_timesteps = sorted(list(use_timesteps))
    return sac, somac


SCHED_KEYS = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
              "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
              "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"]


def gen_schedule(model):
    out = {}
    for S in (4, 50):
        sac, somac = respace(model, S)
        for k in SCHED_KEYS:
            out[f"S{S}_{k}"] = getattr(model, k).clone()
        out[f"S{S}_ori_timesteps"] = np.array(model.ori_timesteps, dtype=np.int64)
        out[f"S{S}_full_sqrt_alphas_cumprod"] = sac
        out[f"S{S}_full_sqrt_one_minus_alphas_cumprod"] = somac
    x0 = synth.synth_tensor("sched/x0", (T, 4, 8, 8))
    nz = synth.synth_tensor("sched/noise", (T, 4, 8, 8))
    t = torch.tensor([999] * T).long()
    out["qs_x0"], out["qs_noise"] = x0, nz
    out["qs_out"] = model.q_sample_respace(x_start=x0, t=t, sqrt_alphas_cumprod=sac, sqrt_one_minus_alphas_cumprod=somac, noise=nz)
    save("g_schedule", **out)


def gen_unet(model):
    unet = model.model.diffusion_model
    sc = model.structcond_stage_model
    h = w = 16
    x = synth.synth_tensor("unet/x", (T, 4, h, w))
    lat = synth.synth_tensor("unet/lat", (T, 4, h, w), 0.5)
    t = torch.tensor([541] * T).long()
    ctx = synth.synth_tensor("ctx", (1, 77, UNET_SMALL["context_dim"]))
    scd = sc(lat, t)
    eps = unet(x, t, context=ctx, struct_cond=scd)
    from ldm.modules.diffusionmodules.util import timestep_embedding
    temb = timestep_embedding(torch.tensor([0, 20, 541, 999]).long(), 64)
    save("g_unet", x=x, lat=lat, t=t.numpy(), ctx=ctx, eps=eps, temb=temb, unet_params=names_shapes(unet),
         struct_params=names_shapes(sc), **{f"sc_{k}": v for k, v in scd.items()})


def gen_vae():
    ae = ref_import.ref("ldm.models.autoencoder")
    cf = ref_import.ref("scripts.wavelet_color_fix")
    vq = ae.VideoAutoencoderKLResi(ddconfig=dict(VAE_DD_SMALL), lossconfig={"target": "torch.nn.Identity"}, embed_dim=4,
                                   fusion_w=1.0, freeze_dec=True, version=1)
    vq.eval()
    synth.fill_module_(vq, "vae")
    x = synth.synth_tensor("vae/x", (T, 3, 64, 64), 0.5)
    post, fea = vq.encode(x)
    z = synth.synth_tensor("vae/z", (T, 4, 8, 8))
    dec = vq.decode(z, fea)
    vq.decoder.fusion_w = 0.5
    dec_w05 = vq.decode(z, fea)
    style = synth.synth_tensor("vae/style", (T, 3, 64, 64), 0.3) + 0.1
    adain = cf.adaptive_instance_normalization(dec, style)
    wav = cf.wavelet_reconstruction(dec, style)
    save("g_vae", x=x, mean=post.mean, logvar=post.logvar, fea0=fea[0], fea1=fea[1], z=z, dec=dec, dec_w05=dec_w05,
         style=style, adain=adain, wavelet=wav, vae_params=names_shapes(vq))


def gen_first_stage(model):
    """AutoencoderKL.encode (first_stage_model) moments for the init latent path (ddpm.py:3906-3943, 3382-3389)."""
    x = synth.synth_tensor("vae/x", (T, 3, 64, 64), 0.5)
    post = model.first_stage_model.encode(x)
    save("g_first_stage", x=x, mean=post.mean, logvar=post.logvar, params=names_shapes(model.first_stage_model))


def gen_sample(model, ddpm):
    S = 4
    uf = ref_import.ref("scripts.util_flow")
    out = {}
    for tag, (h, w), canvas in [("plain", (16, 16), False), ("canvas", (24, 24), True)]:
        respace(model, S)
        ctx = model.cond_stage_model([""])
        lat = synth.synth_tensor(f"sample/{tag}/lat", (T, 4, h, w), 0.5)
        xT = synth.synth_tensor(f"sample/{tag}/xT", (T, 4, h, w))
        noises = [synth.synth_tensor(f"sample/{tag}/noise{i}", (T, 4, h, w)) for i in range(S)]
        ff = synth.smooth_flow(f"sample/{tag}/ff", T - 1, h, w)
        fb = synth.smooth_flow(f"sample/{tag}/fb", T - 1, h, w)
        focc, bocc = uf.forward_backward_consistency_check(fb, ff)
        queue = list(noises)
        orig = ddpm.noise_like
        ddpm.noise_like = lambda shape, device, repeat=False: queue.pop(0)
        try:
            kw = dict(cond=ctx, struct_cond=lat, guidance_scale=-10.0, lr_images=None, flows=(ff[None], fb[None]),
                      masks=(focc[None, :, None], bocc[None, :, None]), batch_size=1, timesteps=S, time_replace=S, x_T=xT,
                      return_intermediates=True, verbose=False)
            if canvas:
                x0, inter = model.sample_canvas(tile_size=16, tile_overlap=8, batch_size_sample=1, **kw)
            else:
                x0, inter = model.sample(**kw)
            # and once more without guidance
            queue[:] = list(noises)
            kw["flows"], kw["masks"] = None, None
            if canvas:
                x0_ng, _ = model.sample_canvas(tile_size=16, tile_overlap=8, batch_size_sample=1, **kw)
            else:
                x0_ng, _ = model.sample(**kw)
        finally:
            ddpm.noise_like = orig
        out.update({f"{tag}_ctx": ctx, f"{tag}_lat": lat, f"{tag}_xT": xT, f"{tag}_noise": torch.stack(noises),
                    f"{tag}_ff": ff, f"{tag}_fb": fb, f"{tag}_focc": focc, f"{tag}_bocc": bocc, f"{tag}_x0": x0,
                    f"{tag}_x0_noguid": x0_ng})
    out["gauss16"] = model._gaussian_weights(16, 16, 1)[0, 0]
    out["gauss64"] = model._gaussian_weights(64, 64, 1)[0, 0]
    save("g_sample", **out)


def gen_sample_lr():
    """The `lr_images` guidance term of p_sample (ddpm.py:4359-4366 -> compute_temporal_condition_v2 :3469-3500): the reference resizes the LR
    frames to the latent grid (bicubic), estimates flows on them with ITS RAFT_SR at every step and pulls the latents along those flows
    (no occlusion masks).  Reduced model with the reference's real flow network (synthetic weights), T frames on a 128 x 128 latent (the
    smallest grid on which the reference's RAFT works: at 64 x 64 the coarsest level of its correlation pyramid is 1 x 1 and its bilinear
    sampler divides by W - 1 = 0 -> NaN flows), 4 steps: once with lr_images alone, once with lr_images AND flows / masks (both terms, in the reference's order)."""
    uf = ref_import.ref("scripts.util_flow")
    model, ddpm = build_ref_model(flownet_config={"target": "basicsr.archs.raft_arch.RAFT_SR", "params": {"model": "normal", "load_path": None}})
    synth.fill_module_(model.flownet_model, "raft")
    from cases import sample_lr_inputs
    c = sample_lr_inputs(T)
    S, h, w, lat, xT, noises, lr, ff, fb = c["S"], c["h"], c["w"], c["lat"], c["xT"], c["noises"], c["lr"], c["ff"], c["fb"]
    respace(model, S)
    ctx = model.cond_stage_model([""])
    focc, bocc = uf.forward_backward_consistency_check(fb, ff)
    out = dict(focc=focc, bocc=bocc)        # (inputs are regenerated from the recipe in cases.py: the fixture holds the reference's outputs)
    orig = ddpm.noise_like
    try:
        for tag, fl, mk in (("lr", None, None), ("lr_flows", (ff[None], fb[None]), (focc[None, :, None], bocc[None, :, None]))):
            queue = list(noises)
            ddpm.noise_like = lambda shape, device, repeat=False: queue.pop(0)
            x0, _ = model.sample(cond=ctx, struct_cond=lat, guidance_scale=-10.0, lr_images=lr, flows=fl, masks=mk, batch_size=1, timesteps=S,
                                 time_replace=S, x_T=xT, return_intermediates=True, verbose=False)
            out[f"x0_{tag}"] = x0
    finally:
        ddpm.noise_like = orig
    with torch.no_grad():       # the flows the guidance used (for the tests' diagnostics)
        res = torch.nn.functional.interpolate(lr, size=(h, w), mode="bicubic")
        f_f, f_b = model.compute_flow(res[None])
    assert bool(torch.isfinite(f_f).all()) and bool(torch.isfinite(out["x0_lr"]).all())
    out.update(lr_flow_f=f_f[0], lr_flow_b=f_b[0], raft_names_shapes=names_shapes(model.flownet_model))
    save("g_sample_lr", **out)


def gen_pstep(model, ddpm):
    """The single-step API (ddpm.py:4157-4189, 4325-4380, 4191-4322, 4383-4442) and decode_first_stage (:3786) of the reference on the
    reduced model: one p_mean_variance / p_sample (guided) at schedule index 2 of a 4-step schedule, one p_sample_canvas on a 24x24
    latent with 16/8 tiles, and first_stage_model.decode through decode_first_stage."""
    uf = ref_import.ref("scripts.util_flow")
    S, i = 4, 2
    respace(model, S)
    ctx = model.cond_stage_model([""])
    out = {}
    for tag, (h, w) in (("plain", (16, 16)), ("canvas", (24, 24))):
        x = synth.synth_tensor(f"pstep/{tag}/x", (T, 4, h, w))
        lat = synth.synth_tensor(f"pstep/{tag}/lat", (T, 4, h, w), 0.5)
        nz = synth.synth_tensor(f"pstep/{tag}/noise", (T, 4, h, w))
        ff, fb = synth.smooth_flow(f"pstep/{tag}/ff", T - 1, h, w), synth.smooth_flow(f"pstep/{tag}/fb", T - 1, h, w)
        focc, bocc = uf.forward_backward_consistency_check(fb, ff)
        flows, masks = (ff[None], fb[None]), (focc[None, :, None], bocc[None, :, None])
        ts = torch.full((1,), i, dtype=torch.long)
        t_rep = torch.tensor([model.ori_timesteps[i]] * T).long()
        orig = ddpm.noise_like
        ddpm.noise_like = lambda shape, device, repeat=False: nz
        try:
            if tag == "plain":
                sc = model.structcond_stage_model(lat, t_rep)
                mean, var, logvar, x0 = model.p_mean_variance(x=x, c=ctx, struct_cond=sc, t=ts, clip_denoised=False, return_x0=True, t_replace=t_rep)
                z = model.p_sample(x, ctx, sc, ts, guidance_scale=-10.0, flows=flows, masks=masks, t_replace=t_rep)
            else:
                tw = model._gaussian_weights(16, 16, 1)
                mean, var, logvar, x0 = model.p_mean_variance_canvas(x=x, c=ctx, struct_cond=lat, t=ts, clip_denoised=False, return_x0=True,
                                                                     t_replace=t_rep[:1], tile_size=16, tile_overlap=8, batch_size=1, tile_weights=tw)
                z = model.p_sample_canvas(x, ctx, lat, ts, guidance_scale=-10.0, flows=flows, masks=masks, t_replace=t_rep[:1], tile_size=16,
                                          tile_overlap=8, batch_size=1, tile_weights=tw)
        finally:
            ddpm.noise_like = orig
        out.update({f"{tag}_x": x, f"{tag}_lat": lat, f"{tag}_noise": nz, f"{tag}_ff": ff, f"{tag}_fb": fb, f"{tag}_focc": focc, f"{tag}_bocc": bocc,
                    f"{tag}_mean": mean, f"{tag}_var": var, f"{tag}_logvar": logvar, f"{tag}_x0": x0, f"{tag}_z": z})
    zl = synth.synth_tensor("pstep/dec/z", (T, 4, 8, 8))
    out["dec_z"], out["dec_out"] = zl, model.decode_first_stage(zl)
    out["ctx"] = ctx
    save("g_pstep", **out)


def gen_fullwidth():
    """G10: BASELINE configs[0] at FULL width through the reference's own classes — one 512x512 frame (T = 1,
    latent 64x64), 4 DDPM steps, no flows: first-stage encode -> q_sample_respace -> model.sample -> video-VAE encode /
    decode -> AdaIN -> clamp, the call sequence of the script's per-segment body (oldcanvas_tile.py:429-471 without the tile
    arguments).  Latents are stored in full (64 KiB each), the 3 MB pixel tensors as stride-4 slices + norms; inputs and
    weights are regenerated on the test side from the same synth recipes.  ~2 min on 8 CPU threads: not part of the default
    `make_golden.py` run (`make_golden.py fullwidth`)."""
    from configs import STRUCT_FULL, UNET_FULL, VAE_DD_FULL
    ae = ref_import.ref("ldm.models.autoencoder")
    cf = ref_import.ref("scripts.wavelet_color_fix")
    Tn, S, H, h = 1, 4, 512, 64
    ucfg, scfg, dd = dict(UNET_FULL, num_frames=Tn), dict(STRUCT_FULL, num_frames=Tn), dict(VAE_DD_FULL, num_frames=Tn)
    model, ddpm = build_ref_model(ucfg, scfg, dd, Tn)
    vq = ae.VideoAutoencoderKLResi(ddconfig=dd, lossconfig={"target": "torch.nn.Identity"}, embed_dim=4, fusion_w=1.0,
                                   freeze_dec=True, version=1).eval()
    synth.fill_module_(vq, "vae")
    x = synth.synth_tensor("one/x", (Tn, 3, H, H), 0.5).clamp(-1, 1)
    n_post, n0 = synth.synth_tensor("one/np", (Tn, 4, h, h)), synth.synth_tensor("one/n0", (Tn, 4, h, h))
    steps = [synth.synth_tensor(f"one/n{i}", (Tn, 4, h, h)) for i in range(S)]          # indexed by schedule index i
    sac, somac = respace(model, S)
    post = model.first_stage_model.encode(x)
    init = model.scale_factor * (post.mean + post.std * n_post)                          # get_first_stage_encoding with injected noise
    ctx = model.cond_stage_model([""])
    xT = model.q_sample_respace(x_start=init, t=torch.full((Tn,), 999).long(), sqrt_alphas_cumprod=sac,
                                sqrt_one_minus_alphas_cumprod=somac, noise=n0)
    # one network evaluation at the first step's timestep (per-network full-width parity against the reference itself)
    t0 = torch.tensor([model.ori_timesteps[S - 1]] * Tn).long()
    scd = model.structcond_stage_model(init, t0)
    eps0 = model.model.diffusion_model(xT, t0, context=ctx, struct_cond=scd)
    queue = [steps[i] for i in reversed(range(S))]                                       # loop order i = S-1 .. 0
    orig = ddpm.noise_like
    ddpm.noise_like = lambda shape, device, repeat=False: queue.pop(0)
    try:
        x0, _ = model.sample(cond=ctx, struct_cond=init, guidance_scale=-10.0, lr_images=None, flows=None, masks=None,
                             batch_size=1, timesteps=S, time_replace=S, x_T=xT, return_intermediates=True, verbose=False)
    finally:
        ddpm.noise_like = orig
    _, fea = vq.encode(x)
    dec = vq.decode(x0 * 1. / model.scale_factor, fea)
    out = torch.clamp((cf.adaptive_instance_normalization(dec, x) + 1.0) / 2.0, min=0.0, max=1.0)
    nrm = lambda t: np.array([float(t.double().norm()), float(t.double().mean())])
    save("g_full_c1", init=init, xT=xT, eps0=eps0, x0=x0, dec_s4=dec[:, :, ::4, ::4], dec_norm=nrm(dec), out_s4=out[:, :, ::4, ::4],
         out_norm=nrm(out), fea0_s8=fea[0][:, ::8, ::8, ::8], fea0_norm=nrm(fea[0]), fea1_s4=fea[1][:, ::8, ::4, ::4],
         fea1_norm=nrm(fea[1]), post_mean=post.mean, post_logvar=post.logvar,
         **{f"sc_{k}_norm": nrm(v) for k, v in scd.items()}, sc_8=scd["8"])


def gen_workload(name="c2", S=4):
    """BASELINE configs[1]-[4] at FULL width through the reference's own classes (tests/golden/cases.py lists the workloads): the
    per-segment body of the script (oldcanvas_tile.py:429-471) — first-stage encode -> q_sample_respace -> model.sample /
    sample_canvas -> video-VAE encode / decode (dec_w = 1 and the script's default 0.5) -> AdaIN -> clamp.  Latents are stored in
    full, pixel tensors as strided slices + norms.  CPU minutes on 8 threads: c2 S=4 ~6, c2g S=4 ~6, c4 S=4 ~25, c2 S=50 ~40; not
    part of the default run (`make_golden.py workload:c2:4`)."""
    from cases import case_inputs
    from configs import STRUCT_FULL, UNET_FULL, VAE_DD_FULL
    ae = ref_import.ref("ldm.models.autoencoder")
    cf = ref_import.ref("scripts.wavelet_color_fix")
    uf = ref_import.ref("scripts.util_flow")
    S = int(S)
    c = case_inputs(name, S)
    Tn, H, h, x, noise, st = c["T"], c["H"], c["h"], c["x"], c["noise"], c["stride"]
    ucfg, scfg, dd = dict(UNET_FULL, num_frames=Tn), dict(STRUCT_FULL, num_frames=Tn), dict(VAE_DD_FULL, num_frames=Tn)
    model, ddpm = build_ref_model(ucfg, scfg, dd, Tn)
    vq = ae.VideoAutoencoderKLResi(ddconfig=dd, lossconfig={"target": "torch.nn.Identity"}, embed_dim=4, fusion_w=1.0,
                                   freeze_dec=True, version=1).eval()
    synth.fill_module_(vq, "vae")
    sac, somac = respace(model, S)
    post = model.first_stage_model.encode(x)
    init = model.scale_factor * (post.mean + post.std * noise["posterior"])
    ctx = model.cond_stage_model([""])
    xT = model.q_sample_respace(x_start=init, t=torch.full((Tn,), 999).long(), sqrt_alphas_cumprod=sac,
                                sqrt_one_minus_alphas_cumprod=somac, noise=noise["x_T"])
    flows = masks = None
    extra = {}
    if c["ff"] is not None:
        focc, bocc = uf.forward_backward_consistency_check(c["fb"], c["ff"])
        flows, masks = (c["ff"][None], c["fb"][None]), (focc[None, :, None], bocc[None, :, None])
        extra.update(focc=focc, bocc=bocc)
    queue = [noise["steps"][i] for i in reversed(range(S))]                                # loop order i = S-1 .. 0
    orig = ddpm.noise_like
    ddpm.noise_like = lambda shape, device, repeat=False: queue.pop(0)
    try:
        kw = dict(cond=ctx, struct_cond=init, guidance_scale=-10.0, lr_images=None, flows=flows, masks=masks, batch_size=1,
                  timesteps=S, time_replace=S, x_T=xT, return_intermediates=True, verbose=False)
        if c["canvas"]:
            x0, _ = model.sample_canvas(tile_size=64, tile_overlap=32, batch_size_sample=1, **kw)
        else:
            x0, _ = model.sample(**kw)
    finally:
        ddpm.noise_like = orig
    print("sampled", flush=True)
    del model
    _, fea = vq.encode(x)
    nrm = lambda t: np.array([float(t.double().norm()), float(t.double().mean())])
    for w, sfx in ((1.0, ""), (0.5, "_w05")):
        vq.decoder.fusion_w = w
        dec = vq.decode(x0 * 1. / 0.18215, fea)
        out = torch.clamp((cf.adaptive_instance_normalization(dec, x) + 1.0) / 2.0, min=0.0, max=1.0)
        extra.update({f"dec{sfx}_s": dec[:, :, ::st, ::st].clone(), f"dec{sfx}_norm": nrm(dec), f"out{sfx}_s": out[:, :, ::st, ::st].clone(),
                      f"out{sfx}_norm": nrm(out)})
        del dec, out
    save(f"g_work_{name}_S{S}", init=init, xT=xT, x0=x0, fea0_norm=nrm(fea[0]), fea1_norm=nrm(fea[1]), **extra)


def gen_sample_opts(model=None, ddpm=None):
    """The option branches of p_sample_loop that sit between steps (ddpm.py:4501-4599) on the reduced model, 4-step schedule, unguided:
    start_T (skip the indices whose original timestep is above it), mask + x0 (inpainting blend with q_sample(x0, ts): its randn_like
    draws come from the global generator, seeded here and regenerated on the test side), adain_fea (latent-space AdaIN after the
    last step), callback / img_callback (call order + the images passed)."""
    if model is None:
        model, ddpm = build_ref_model()
    S, h, w = 4, 16, 16
    respace(model, S)
    ctx = model.cond_stage_model([""])
    lat = synth.synth_tensor("opts/lat", (T, 4, h, w), 0.5)
    xT = synth.synth_tensor("opts/xT", (T, 4, h, w))
    noises = [synth.synth_tensor(f"opts/noise{i}", (T, 4, h, w)) for i in range(S)]          # indexed by schedule index i
    x0m = synth.synth_tensor("opts/x0", (T, 4, h, w), 0.7)
    mask = (synth.synth_tensor("opts/mask", (T, 1, h, w)) > 0.3).float()
    fea = synth.synth_tensor("opts/adain", (T, 4, h, w), 0.4) + 0.2
    out = {"lat": lat, "xT": xT, "noise": torch.stack(noises), "x0m": x0m, "mask": mask, "adain_fea": fea, "ctx": ctx,
           "ori_timesteps": np.array(model.ori_timesteps, dtype=np.int64)}
    orig = ddpm.noise_like

    def run(**opt):
        # the loop draws noise_like once per EXECUTED step, in loop order i = S-1 .. 0 (skipped steps draw nothing)
        st = opt.get("start_T")
        order = [i for i in reversed(range(S)) if st is None or model.ori_timesteps[i] <= st]
        queue = [noises[i] for i in order]
        ddpm.noise_like = lambda shape, device, repeat=False: queue.pop(0)
        torch.manual_seed(4242)
        try:
            return model.p_sample_loop(ctx, lat, (T, 4, h, w), guidance_scale=-10.0, x_T=xT, verbose=False, timesteps=S, time_replace=S,
                                       **opt)
        finally:
            ddpm.noise_like = orig
    out["x_start_T"] = run(start_T=600)
    out["x_mask"] = run(mask=mask, x0=x0m)
    out["x_adain"] = run(adain_fea=fea)
    calls, imgs = [], []
    out["x_cb"] = run(callback=lambda i: calls.append(("cb", i)), img_callback=lambda img, i: (calls.append(("img", i)), imgs.append(img.clone())))
    out["cb_order"] = np.array([[0 if k == "cb" else 1, i] for k, i in calls], dtype=np.int64)
    out["cb_imgs"] = torch.stack(imgs)
    save("g_sample_opts", **out)


def gen_sample_opts_canvas():
    """start_T on the CANVAS loop (advisor item of round 3): p_sample_loop_canvas has another rule than p_sample_loop — `timesteps =
    min(timesteps, start_T)` (ddpm.py:4639-4640), i.e. it walks the schedule indices start_T-1 .. 0 whatever their original timesteps
    are.  Reduced model, 4-step schedule, 24x24 latent with 16/8 tiles, start_T = 3 -> indices 2, 1, 0.  -> g_sample_opts_canvas.npz"""
    model, ddpm = build_ref_model()
    S, h, w, st = 4, 24, 24, 3
    respace(model, S)
    ctx = model.cond_stage_model([""])
    lat = synth.synth_tensor("optsc/lat", (T, 4, h, w), 0.5)
    xT = synth.synth_tensor("optsc/xT", (T, 4, h, w))
    noises = [synth.synth_tensor(f"optsc/noise{i}", (T, 4, h, w)) for i in range(S)]          # indexed by schedule index i
    queue = [noises[i] for i in reversed(range(min(S, st)))]
    orig = ddpm.noise_like
    ddpm.noise_like = lambda shape, device, repeat=False: queue.pop(0)
    try:
        x = model.p_sample_loop_canvas(ctx, lat, (T, 4, h, w), guidance_scale=-10.0, x_T=xT, verbose=False, timesteps=S, time_replace=S,
                                       start_T=st, tile_size=16, tile_overlap=8, batch_size=1)
    finally:
        ddpm.noise_like = orig
    assert not queue
    save("g_sample_opts_canvas", lat=lat, xT=xT, noise=torch.stack(noises), ctx=ctx, x_start_T=x, start_T=np.array([st]),
         ori_timesteps=np.array(model.ori_timesteps, dtype=np.int64))


def gen_ckpt_keys():
    """The key / shape list of the checkpoints the scripts load (oldcanvas_tile.py:91-108, :296-306): `state_dict` of the FULL-width
    LatentDiffusionVSRTextWT exactly as the shipped YAML builds it (mgldvsr_512_realbasicvsr_deg.yaml: UNet, struct-cond encoder,
    first-stage KL-VAE, RAFT_SR flow net, 1000-step schedule buffers) and of the video VAE (video_autoencoder_kl_64x64x4_resi.yaml), read
    from the reference's own classes.  The `cond_stage_model.*` entries cannot come from the reference here — FrozenOpenCLIPEmbedder needs
    open_clip, which is not installed — so that part of the list is open_clip's published ViT-H-14 TEXT tower layout (width 1024, 24
    layers, vocabulary 49408, context 77; `visual` is deleted by the embedder, modules.py:153) and is marked as such in the fixture.
    Names and shapes only (JSON, ~300 KB): the test builds an all-keys state dict from it and loads it through VSRPipeline."""
    from configs import STRUCT_FULL, UNET_FULL, VAE_DD_FULL
    ae = ref_import.ref("ldm.models.autoencoder")
    model, _ = build_ref_model(dict(UNET_FULL), dict(STRUCT_FULL), dict(VAE_DD_FULL), 5,
                               flownet_config={"target": "basicsr.archs.raft_arch.RAFT_SR", "params": {"model": "normal", "load_path": None}})
    ldm_keys = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in model.state_dict().items()]
    vq = ae.VideoAutoencoderKLResi(ddconfig=dict(VAE_DD_FULL), lossconfig={"target": "torch.nn.Identity"}, embed_dim=4, fusion_w=1.0,
                                   freeze_dec=True, version=1)
    vq_keys = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in vq.state_dict().items()]
    W, L, V, C = 1024, 24, 49408, 77
    tk = "cond_stage_model.model."
    clip = [[tk + "positional_embedding", [C, W]], [tk + "text_projection", [W, W]], [tk + "logit_scale", []],
            [tk + "token_embedding.weight", [V, W]]]
    for i in range(L):
        b = f"{tk}transformer.resblocks.{i}."
        clip += [[b + "ln_1.weight", [W]], [b + "ln_1.bias", [W]], [b + "attn.in_proj_weight", [3 * W, W]], [b + "attn.in_proj_bias", [3 * W]],
                 [b + "attn.out_proj.weight", [W, W]], [b + "attn.out_proj.bias", [W]], [b + "ln_2.weight", [W]], [b + "ln_2.bias", [W]],
                 [b + "mlp.c_fc.weight", [4 * W, W]], [b + "mlp.c_fc.bias", [4 * W]], [b + "mlp.c_proj.weight", [W, 4 * W]],
                 [b + "mlp.c_proj.bias", [W]]]
    clip += [[tk + "ln_final.weight", [W]], [tk + "ln_final.bias", [W]]]
    path = os.path.join(OUT, "g_ckpt_keys.json")
    with open(path, "w") as fh:
        json.dump({"ldm": ldm_keys, "vq": vq_keys, "cond_stage_openclip_vit_h_14_text": [k + ["float32"] for k in clip],
                   "note": "ldm / vq: read from the reference classes; cond_stage: open_clip's published text-tower layout (unpinned: open_clip not installed)"},
                  fh)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB), {len(ldm_keys)} + {len(vq_keys)} + {len(clip)} keys")


class _AD(dict):
    """dict with attribute access (the scripts read `config.model`, instantiate_from_config reads it as a dict)"""
    __getattr__ = dict.__getitem__

    @staticmethod
    def wrap(v):
        return _AD({k: _AD.wrap(x) for k, x in v.items()}) if isinstance(v, dict) else v


def _harness_env(tmp, resolution, full=False, frames=T):
    """What a run of one of the reference's entry scripts needs here beyond ref_import's stubs: reduced-width (or, full = True, the
    shipped full-width) configs behind OmegaConf.load, synthetic checkpoints in the reference's key layout, Module.cuda as a no-op.
    Returns the ddpm module."""
    ref_import.install()
    stubs = types.ModuleType("golden_stubs")
    stubs.StubCond, stubs.StubFlow = _StubCond, _StubFlow
    sys.modules["golden_stubs"] = stubs
    if full:
        from configs import STRUCT_FULL, UNET_FULL, VAE_DD_FULL
        UNET_SMALL, STRUCT_SMALL, VAE_DD_SMALL = (dict(UNET_FULL, num_frames=frames), dict(STRUCT_FULL, num_frames=frames),
                                                  dict(VAE_DD_FULL, num_frames=frames))
    else:
        from configs import STRUCT_SMALL, UNET_SMALL, VAE_DD_SMALL
    T = frames
    fs_dd = dict(VAE_DD_SMALL, resolution=resolution)
    fs_dd.pop("num_frames")
    dcfg = {"target": "ldm.models.diffusion.ddpm.LatentDiffusionVSRTextWT", "params": dict(
        first_stage_config={"target": "ldm.models.autoencoder.AutoencoderKL",
                            "params": {"embed_dim": 4, "ddconfig": fs_dd, "lossconfig": {"target": "torch.nn.Identity"}}},
        cond_stage_config={"target": "golden_stubs.StubCond", "params": {"ctx_dim": UNET_SMALL["context_dim"]}},
        structcond_stage_config={"target": "ldm.modules.diffusionmodules.openaimodel.InflatedEncoderUNetModelWT", "params": dict(STRUCT_SMALL)},
        flownet_config={"target": "basicsr.archs.raft_arch.RAFT_SR", "params": {"model": "normal", "load_path": None}},
        num_frames=T, linear_start=0.00085, linear_end=0.0120, num_timesteps_cond=1, log_every_t=200, timesteps=1000,
        first_stage_key="image", cond_stage_key="caption", image_size=resolution, channels=4, cond_stage_trainable=False,
        conditioning_key="crossattn", scale_factor=0.18215, use_ema=False, time_replace=1000, use_usm=True,
        unet_config={"target": "ldm.modules.diffusionmodules.openaimodel.InflatedUNetModelDualcondV2", "params": dict(UNET_SMALL)})}
    vcfg = {"target": "ldm.models.autoencoder.VideoAutoencoderKLResi", "params": dict(
        embed_dim=4, fusion_w=1.0, freeze_dec=True, synthesis_data=False, version=1, lossconfig={"target": "torch.nn.Identity"},
        ddconfig=dict(VAE_DD_SMALL, resolution=resolution))}

    class OmegaConf:
        @staticmethod
        def load(path):
            return _AD.wrap({"model": vcfg if "video_autoencoder" in str(path) or "video_vae" in str(path) else dcfg})
    sys.modules["omegaconf"].OmegaConf = OmegaConf
    if os.path.join(ref_import.REF, "scripts") not in sys.path:
        sys.path.insert(0, os.path.join(ref_import.REF, "scripts"))        # `from util_image import ImageSpliterTh`
    ddpm = ref_import.ref("ldm.models.diffusion.ddpm")
    util = ref_import.ref("ldm.util")
    model = util.instantiate_from_config(dcfg)
    for mod, salt in ((model.model.diffusion_model, "unet"), (model.structcond_stage_model, "structcond"),
                      (model.first_stage_model, "first_stage"), (model.flownet_model, "raft")):
        synth.fill_module_(mod, salt)
    torch.save({"state_dict": model.state_dict()}, os.path.join(tmp, "model.ckpt"))
    vq = util.instantiate_from_config(vcfg)
    synth.fill_module_(vq, "vae")
    torch.save({"state_dict": vq.state_dict()}, os.path.join(tmp, "vqgan.ckpt"))
    return ddpm


def _harness_frames(tmp, salt, h, w, n):
    """n smooth, textured, slowly translating LR frames [n,h,w,3] uint8 (so that RAFT has something to match), written as PNGs"""
    from PIL import Image
    base = torch.nn.functional.avg_pool2d(torch.sigmoid(synth.synth_tensor(salt, (1, 3, h + 16, w + 16), 1.8)), 5, 1, 2)
    lr = torch.stack([base[0, :, 4 + 2 * k:4 + 2 * k + h, 6 + k:6 + k + w] for k in range(n)])
    lr_u8 = (lr.clamp(0, 1) * 255).round().byte().permute(0, 2, 3, 1).numpy()
    os.makedirs(os.path.join(tmp, "in", "seq0"))
    for k in range(n):
        Image.fromarray(lr_u8[k]).save(os.path.join(tmp, "in", "seq0", f"{k:04d}.png"))
    return lr_u8


class _Instrument:
    """record every torch.randn / randn_like draw and the arguments / result of the sampler entry a script calls"""

    def __init__(self, ddpm, entry):
        self.ddpm, self.entry, self.draws, self.calls = ddpm, entry, [], []

    def __enter__(self):
        self.o = (torch.randn, torch.randn_like, getattr(self.ddpm.LatentDiffusionVSRTextWT, self.entry), torch.nn.Module.cuda, sys.argv)
        o_randn, o_like, o_entry = self.o[:3]

        def randn(*a, **k):
            t = o_randn(*a, **k)
            self.draws.append(t.clone())
            return t

        def randn_like(x, **k):
            t = o_like(x, **k)
            self.draws.append(t.clone())
            return t

        def entry(obj, **kw):
            rng = torch.get_rng_state().clone()          # CPU generator at entry: the per-step draws follow from it
            out = o_entry(obj, **kw)
            self.calls.append({"rng": rng, "x_T": kw["x_T"].clone(), "ff": kw["flows"][0].clone(), "fb": kw["flows"][1].clone(),
                               "fo": kw["masks"][0].clone(), "bo": kw["masks"][1].clone(), "lat": kw["struct_cond"].clone(),
                               "x0": out[0].clone(), "gscale": float(kw["guidance_scale"])})
            return out
        torch.randn, torch.randn_like = randn, randn_like
        setattr(self.ddpm.LatentDiffusionVSRTextWT, self.entry, entry)
        torch.nn.Module.cuda = lambda m, *a, **k: m
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like = self.o[0], self.o[1]
        setattr(self.ddpm.LatentDiffusionVSRTextWT, self.entry, self.o[2])
        torch.nn.Module.cuda, sys.argv = self.o[3], self.o[4]
        return False


def _steps_digest(state, draws):
    """the per-step noise draws of a sampler call as (CPU generator state at its entry, sha256 of the fp32 draws, a few head values):
    50 x [T,4,64,64] floats are 16 MiB a call; they are torch.randn calls on the global generator, so the state reproduces them (checked
    here) on the torch build that made the fixture, and the tests check the digest before they trust the replay"""
    import hashlib
    steps = torch.stack(draws)
    keep = torch.get_rng_state()
    torch.set_rng_state(state)
    rep = torch.stack([torch.randn(tuple(steps.shape[1:])) for _ in range(steps.shape[0])])
    torch.set_rng_state(keep)
    assert torch.equal(rep, steps), "the generator state at the sampler's entry does not reproduce its per-step draws"
    return state, np.frombuffer(hashlib.sha256(steps.numpy().tobytes()).digest(), dtype=np.uint8), steps[:, 0, 0, 0, :4]


def gen_harness():
    """H4 (SURVEY 8(a), 8(c) "G9"): the reference's README entry script ITSELF —
    scripts/vsr_val_ddpm_text_T_vqganfin_oldcanvas_tile.py::main(), imported from the reference tree and run unmodified — on a tiny
    PNG sequence, with reduced-width networks, synthetic checkpoints written to a temp dir and the third-party modules it cannot
    import here stubbed (OmegaConf.load -> the reduced configs, Module.cuda -> no-op; the text tower -> a constant context).
    3 LR frames of 136x136 (-> bicubic x4 = 544x544, already a multiple of 32), 2 DDPM steps, --vqgantile_size 512 /
    --vqgantile_stride 32 so that the script takes its large-frame branch (2x2 pixel patches of 512^2 through the whole sampler,
    aggregation sampling with one 64x64 latent tile each; the small-frame branch of the reference raises UnboundLocalError,
    SURVEY 3.1), RAFT flows, dec_w 0.5, AdaIN.  The fixture holds the LR frames, the noise the script drew (identical for every
    patch: it re-seeds per patch), per-patch x_T / flows / masks / x_0 and the uint8 HR frames it wrote.  ~1 min: not part of
    the default run (`make_golden.py harness`)."""
    import shutil
    import tempfile
    from PIL import Image
    tmp = tempfile.mkdtemp(prefix="mgld_harness_")
    Tn, S, LR = T, 2, 136
    ddpm = _harness_env(tmp, 512)
    lr_u8 = _harness_frames(tmp, "harness/img", LR, LR, Tn)
    script = ref_import.ref("scripts.vsr_val_ddpm_text_T_vqganfin_oldcanvas_tile")
    with _Instrument(ddpm, "sample_canvas") as ins:
        sys.argv = ["x", "--seqs-path", os.path.join(tmp, "in"), "--outdir", os.path.join(tmp, "out"), "--ddpm_steps", str(S), "--n_frames",
                    str(Tn), "--config", "diffusion.yaml", "--ckpt", os.path.join(tmp, "model.ckpt"), "--vqgan_ckpt",
                    os.path.join(tmp, "vqgan.ckpt"), "--seed", "42", "--dec_w", "0.5", "--colorfix_type", "adain", "--vqgantile_size", "512",
                    "--vqgantile_stride", "32", "--upscale", "4"]
        with torch.enable_grad():                       # (the generator runs under set_grad_enabled(False); the script manages grad itself)
            script.main()
    draws, calls = ins.draws, ins.calls
    hr = np.stack([np.asarray(Image.open(os.path.join(tmp, "out", "seq0", f"{k:04d}.png")).convert("RGB")) for k in range(Tn)])
    assert len(calls) == 4 and len(draws) == 4 * (2 + S), (len(calls), len(draws))
    per = 2 + S
    for c in range(1, 4):                               # the script re-seeds per patch: every patch sees the same noise
        for j in range(per):
            assert torch.equal(draws[c * per + j], draws[j])
    # (HR frames: every second pixel + per-frame channel means of the full frames; 2.6 MB of uint8 otherwise)
    out = {"lr_u8": lr_u8, "hr_u8_s2": hr[:, ::2, ::2], "hr_mean": hr.reshape(Tn, -1, 3).astype(np.float64).mean(1),
           "hr_shape": np.array(hr.shape), "noise_posterior": draws[0], "noise_xT": draws[1],
           "noise_steps_loop_order": torch.stack(draws[2:2 + S])}
    for c, rec in enumerate(calls):
        for k, v in rec.items():
            if k in ("gscale", "rng"):
                continue
            if c == 0 or k in ("x0",):
                out[f"p{c}_{k}"] = v
            else:
                out[f"p{c}_{k}_norm"] = np.array([float(v.double().norm())])
    save("g_harness", **out)
    shutil.rmtree(tmp, ignore_errors=True)


def gen_harness_full():
    """H4 at the PRODUCTION schedule and width (review item of round 3): the reference's README entry script
    scripts/vsr_val_ddpm_text_T_vqganfin_oldcanvas_tile.py::main(), imported from the reference tree and run unmodified, with the
    SHIPPED full-width networks (synthetic weights), its default 5 frames per segment and 50 DDPM steps, on 5 LR frames of 136x136
    (-> 544x544: 2x2 pixel patches of 512^2, RAFT flows, dec_w 0.5, AdaIN) — the same stubbing as gen_harness.  The fixture holds the
    LR frames, the noise (the per-step draws as generator state + digest, _steps_digest), per-patch flows / masks (norms) / x_T / x_0 and the
    uint8 HR frames the script wrote.  ~80 CPU-minutes on 8 threads: `make_golden.py harness_full`.  (The committed file came from a run
    that stored the step draws rounded to fp16; the generator state was recovered by re-running the script up to the first sampler call
    and checked against those — this function now records it directly.)"""
    import shutil
    import tempfile
    import time
    from PIL import Image
    tmp = tempfile.mkdtemp(prefix="mgld_harness_full_")
    Tn, S, LR = 5, 50, 136
    t0 = time.time()
    ddpm = _harness_env(tmp, 512, full=True, frames=Tn)
    lr_u8 = _harness_frames(tmp, "harness_full/img", LR, LR, Tn)
    script = ref_import.ref("scripts.vsr_val_ddpm_text_T_vqganfin_oldcanvas_tile")
    with _Instrument(ddpm, "sample_canvas") as ins:
        sys.argv = ["x", "--seqs-path", os.path.join(tmp, "in"), "--outdir", os.path.join(tmp, "out"), "--ddpm_steps", str(S), "--n_frames",
                    str(Tn), "--config", "diffusion.yaml", "--ckpt", os.path.join(tmp, "model.ckpt"), "--vqgan_ckpt",
                    os.path.join(tmp, "vqgan.ckpt"), "--seed", "42", "--dec_w", "0.5", "--colorfix_type", "adain", "--vqgantile_size", "512",
                    "--vqgantile_stride", "32", "--upscale", "4"]
        with torch.enable_grad():
            script.main()
    draws, calls = ins.draws, ins.calls
    hr = np.stack([np.asarray(Image.open(os.path.join(tmp, "out", "seq0", f"{k:04d}.png")).convert("RGB")) for k in range(Tn)])
    per = 2 + S
    assert len(calls) == 4 and len(draws) == 4 * per, (len(calls), len(draws))
    for c in range(1, 4):
        for j in range(per):
            assert torch.equal(draws[c * per + j], draws[j])
    st, sha, head = _steps_digest(calls[0]["rng"], draws[2:2 + S])
    out = {"lr_u8": lr_u8, "hr_u8": hr, "noise_posterior": draws[0], "noise_xT": draws[1],
           "rng_state_steps": st, "noise_steps_sha256": sha, "noise_steps_head": head}
    print(f"gen_harness_full: {time.time() - t0:.0f} s")
    for c, rec in enumerate(calls):
        for k, v in rec.items():
            if k in ("gscale", "rng"):
                continue
            if k in ("x0", "x_T", "lat"):
                out[f"p{c}_{k}"] = v
            elif c == 0:
                out[f"p{c}_{k}"] = v.half() if v.dtype == torch.float32 and k in ("ff", "fb") else v
            else:
                out[f"p{c}_{k}_norm"] = np.array([float(v.double().norm())])
    save("g_harness_full", **out)
    shutil.rmtree(tmp, ignore_errors=True)


def gen_harness_old_full():
    """gen_harness_old at the PRODUCTION width and schedule: scripts/vsr_val_ddpm_text_T_vqganfin_old.py::main() with the shipped
    full-width networks (synthetic weights), one 5-frame segment, 50 DDPM steps (-> g_harness_old_full.npz; the per-step draws as
    generator state + digest, _steps_digest).  ~20 CPU-minutes: `make_golden.py harness_old_full`."""
    gen_harness_old(full=True)


def gen_harness_old(full=False):
    """H4, the two fixed-size entry scripts: scripts/vsr_val_ddpm_text_T_vqganfin_old.py::main() and ..._w_latent.py::main() of the
    reference, run unmodified (same stubbing as gen_harness; torchvision's Resize / CenterCrop — a third-party dependency absent
    here — stand in as the tensor code path of torchvision 0.13/0.14, the reference's pin: bilinear, align_corners=False, no
    antialias, smaller edge -> size; torch.cuda.Event / synchronize -> no-ops).  7 frames of 224x160 (Lanczos leaves them as they
    are: both sides are multiples of 32 -> Resize(128) -> 179x128 -> CenterCrop 128), n_frames 3 (the 7th frame is dropped: no
    repeat-last padding in these scripts), 2 DDPM steps, full-resolution RAFT flows resized by 1/8, dec_w 0.5, AdaIN.  (128 px is
    the smallest frame the reference's RAFT handles: at 64 px its 4-level correlation pyramid ends in a 1x1 level whose
    coordinate normalisation divides by W - 1 = 0 and the flows come out NaN.)  `make_golden.py harness_old`."""
    import shutil
    import tempfile
    from PIL import Image
    ref_import.install()
    torchvision = sys.modules["torchvision"]          # ref_import's stand-in module
    Tn, S, NF = (5, 50, 5) if full else (T, 2, 7)
    out = {}

    class Resize:
        def __init__(self, size):
            self.size = size

        def __call__(self, img):
            h, w = img.shape[-2:]
            if w <= h:
                nw, nh = self.size, int(self.size * h / w)
            else:
                nh, nw = self.size, int(self.size * w / h)
            return torch.nn.functional.interpolate(img, size=[nh, nw], mode="bilinear", align_corners=False, antialias=False)

    class CenterCrop:
        def __init__(self, size):
            self.size = size

        def __call__(self, img):
            h, w = img.shape[-2:]
            top, left = int(round((h - self.size) / 2.0)), int(round((w - self.size) / 2.0))
            return img[..., top:top + self.size, left:left + self.size]

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x
    torchvision.transforms.Resize, torchvision.transforms.CenterCrop, torchvision.transforms.Compose = Resize, CenterCrop, Compose

    class _Ev:
        def __init__(self, *a, **k):
            pass

        def record(self):
            pass

        def elapsed_time(self, other):
            return 0.0
    o_ev, o_sync = torch.cuda.Event, torch.cuda.synchronize
    torch.cuda.Event, torch.cuda.synchronize = _Ev, (lambda *a, **k: None)
    try:
        for tag, modname in (("old", "scripts.vsr_val_ddpm_text_T_vqganfin_old"), ("wlat", "scripts.vsr_val_ddpm_text_T_vqganfin_w_latent")):
            if full and tag != "old":
                continue
            tmp = tempfile.mkdtemp(prefix="mgld_harness_old_")
            ddpm = _harness_env(tmp, 128, full=full, frames=Tn)
            lr_u8 = _harness_frames(tmp, "harness_old/img", 160, 224, NF)
            script = ref_import.ref(modname)
            with _Instrument(ddpm, "sample") as ins:
                sys.argv = ["x", "--seqs-path", os.path.join(tmp, "in"), "--outdir", os.path.join(tmp, "out"), "--ddpm_steps", str(S),
                            "--n_frames", str(Tn), "--config", "diffusion.yaml", "--ckpt", os.path.join(tmp, "model.ckpt"), "--vqgan_ckpt",
                            os.path.join(tmp, "vqgan.ckpt"), "--seed", "42", "--dec_w", "0.5", "--colorfix_type", "adain", "--input_size", "128"]
                if tag == "wlat":
                    sys.argv += ["--latent-dir", os.path.join(tmp, "lat")]
                with torch.enable_grad():
                    script.main()
            names = sorted(os.listdir(os.path.join(tmp, "out", "seq0")))
            nseg = NF // Tn
            assert names == [f"{k:04d}.png" for k in range(nseg * Tn)] and len(ins.calls) == nseg, (names, len(ins.calls))
            hr = np.stack([np.asarray(Image.open(os.path.join(tmp, "out", "seq0", f)).convert("RGB")) for f in names])
            per = 2 + S                                  # posterior noise, x_T noise, one draw per step; seeded once: segments differ
            assert len(ins.draws) == nseg * per
            out["lr_u8"] = lr_u8
            assert bool(torch.isfinite(ins.calls[0]["ff"]).all())
            out[f"{tag}_hr_u8"] = hr
            out[f"{tag}_gscale"] = np.array([c["gscale"] for c in ins.calls])
            for sgi, rec in enumerate(ins.calls):
                d = ins.draws[sgi * per:(sgi + 1) * per]
                out[f"{tag}_s{sgi}_noise_posterior"], out[f"{tag}_s{sgi}_noise_xT"] = d[0], d[1]
                if full:
                    (out[f"{tag}_s{sgi}_rng_state_steps"], out[f"{tag}_s{sgi}_noise_steps_sha256"],
                     out[f"{tag}_s{sgi}_noise_steps_head"]) = _steps_digest(rec["rng"], d[2:])
                else:
                    out[f"{tag}_s{sgi}_noise_steps_loop_order"] = torch.stack(d[2:])
                for k in ("x_T", "ff", "fb", "fo", "bo", "lat", "x0"):
                    out[f"{tag}_s{sgi}_{k}"] = rec[k].half() if full and k in ("ff", "fb") else rec[k]
            if tag == "wlat":
                out["wlat_npy"] = np.stack([np.load(os.path.join(tmp, "lat", "seq0", f"{k:04d}.npy")) for k in range(6)])
            shutil.rmtree(tmp, ignore_errors=True)
    finally:
        torch.cuda.Event, torch.cuda.synchronize = o_ev, o_sync
    save("g_harness_old_full" if full else "g_harness_old", **out)


def gen_raft():
    """RAFT_SR ('normal') of the reference on synthetic weights: two 3-frame clips of 40x56 LR frames (padding path of
    InputPadder exercised: 40 is a multiple of 8, 56 is; use 44x60 instead) -> compute_flow-style pairs, 4 iterations."""
    ra = ref_import.ref("basicsr.archs.raft_arch")
    net = ra.RAFT_SR(model="normal").eval()
    synth.fill_module_(net, "raft")
    h, w = 124, 132   # padded to 128x136 by InputPadder (both paddings exercised); 1/8 grid 16x17, pyramid down to 2x2
    lrs = torch.sigmoid(synth.synth_tensor("raft/lr", (1, 3, 3, h, w), 1.5))
    # smooth + textured content so that correlation has structure: low-pass the noise a little
    lrs = torch.nn.functional.avg_pool2d(lrs.view(3, 3, h, w), 3, 1, 1).view(1, 3, 3, h, w)
    a, b = lrs[:, :-1].reshape(-1, 3, h, w), lrs[:, 1:].reshape(-1, 3, h, w)
    out = {}
    for iters in (1, 4):
        out[f"bwd{iters}"] = net(a, b, iters=iters)
        out[f"fwd{iters}"] = net(b, a, iters=iters)
    fmap = net.fnet([a, b])
    save("g_raft", lrs=lrs, fmap1=fmap[0], cnet=net.cnet(a), names_shapes=names_shapes(net), **out)


def gen_text_hf():
    """Text tower pin.  open_clip (FrozenOpenCLIPEmbedder's dependency, modules.py:12) is not installed here, but transformers'
    CLIPTextModel — the class the reference's FrozenCLIPEmbedder binds (modules.py:7, :207) and an independent published
    implementation of the same text tower (token + position embedding, pre-LN causal blocks, exact GELU, final LayerNorm) —
    is: a tiny random-weight instance gives the expected `last` and `penultimate` (hidden_states[-2] -> final_layer_norm,
    the FrozenOpenCLIPEmbedder layer='penultimate' semantics, modules.py:181-199) outputs."""
    from transformers import CLIPTextConfig, CLIPTextModel
    cfg = CLIPTextConfig(vocab_size=100, hidden_size=64, intermediate_size=256, num_hidden_layers=3, num_attention_heads=4,
                         max_position_embeddings=77, hidden_act="gelu", bos_token_id=98, eos_token_id=99, pad_token_id=0)
    m = CLIPTextModel(cfg).eval()
    synth.fill_module_(m, "text_hf")
    tokens = torch.zeros(2, 77, dtype=torch.long)
    tokens[0, 0], tokens[0, 1] = 98, 99                                  # the empty prompt: [SOT, EOT, 0...]
    g = torch.Generator().manual_seed(7)
    tokens[1, 0] = 98
    tokens[1, 1:40] = torch.randint(1, 98, (39,), generator=g)
    tokens[1, 40] = 99
    out = m(input_ids=tokens, output_hidden_states=True)
    pen = m.final_layer_norm(out.hidden_states[-2])          # (transformers 5.x: CLIPTextModel owns the tower directly)
    save("g_text_hf", tokens=tokens, last=out.last_hidden_state, penultimate=pen, names_shapes=names_shapes(m))


def gen_text_openclip():
    """Text tower pin, round 5: the REFERENCE's own FrozenOpenCLIPEmbedder (ldm/modules/encoders/modules.py:140-199) is instantiated and
    its `encode_with_transformer` / `text_transformer_forward` run — the embedding sum, NLD <-> LND permutes, the `attn_mask` it passes,
    the layer slice of layer="penultimate" (`break` at len(resblocks) - layer_idx) and `ln_final` are the reference's code, executed.
    What the reference binds through `open_clip.create_model_and_transforms` is NOT installable here (un-vendored dependency
    open_clip_torch, modules.py:12); the model object handed to the reference class is a stand-in that restates open_clip's published
    text tower attribute for attribute (token_embedding, positional_embedding, transformer.resblocks[i] = ResidualAttentionBlock
    {ln_1, attn = nn.MultiheadAttention, ln_2, mlp.c_fc / gelu / c_proj} with forward x + attn(ln_1 x, mask); x + mlp(ln_2 x),
    ln_final, the additive causal `attn_mask` buffer, transformer.grad_checkpointing) at reduced width.  The fixture holds tokens and the
    reference class's outputs for both layer choices; oracle/text.py and the product's tower are compared against it."""
    import collections
    import types
    import torch.nn as nn

    class ResidualAttentionBlock(nn.Module):
        def __init__(self, d_model, n_head):
            super().__init__()
            self.ln_1 = nn.LayerNorm(d_model)
            self.attn = nn.MultiheadAttention(d_model, n_head)
            self.ln_2 = nn.LayerNorm(d_model)
            self.mlp = nn.Sequential(collections.OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", nn.GELU()),
                                                              ("c_proj", nn.Linear(d_model * 4, d_model))]))

        def forward(self, x, attn_mask=None):
            h = self.ln_1(x)
            x = x + self.attn(h, h, h, need_weights=False, attn_mask=attn_mask)[0]
            return x + self.mlp(self.ln_2(x))

    class Transformer(nn.Module):
        def __init__(self, width, layers, heads):
            super().__init__()
            self.grad_checkpointing = False
            self.resblocks = nn.ModuleList([ResidualAttentionBlock(width, heads) for _ in range(layers)])

    class CLIPStandIn(nn.Module):
        def __init__(self, vocab=512, ctx=77, width=128, layers=4, heads=2):
            super().__init__()
            self.visual = nn.Identity()                                  # (the reference deletes it, modules.py:154)
            self.token_embedding = nn.Embedding(vocab, width)
            self.positional_embedding = nn.Parameter(torch.empty(ctx, width))
            self.transformer = Transformer(width, layers, heads)
            self.ln_final = nn.LayerNorm(width)
            self.text_projection = nn.Parameter(torch.empty(width, width))
            self.logit_scale = nn.Parameter(torch.ones([]))
            mask = torch.empty(ctx, ctx).fill_(float("-inf")).triu_(1)   # open_clip build_attention_mask
            self.register_buffer("attn_mask", mask, persistent=False)

    made = []

    def create_model_and_transforms(arch, device=None, pretrained=None):
        made.append((arch, pretrained))
        return CLIPStandIn(), None, None

    import importlib.machinery
    if "torchvision" in sys.modules and getattr(sys.modules["torchvision"], "__spec__", None) is None:   # the shim's stub: transformers probes find_spec()
        for k in [k for k in sys.modules if k == "torchvision" or k.startswith("torchvision.")]:
            sys.modules[k].__spec__ = importlib.machinery.ModuleSpec(k, None)
    import transformers  # noqa: F401  (modules.py:6-7 imports it; resolve it before the shim stubs torchvision)
    from transformers import CLIPTextModel, CLIPTokenizer  # noqa: F401
    import transformers.models.clip.modeling_clip as hf_clip
    if not hasattr(hf_clip, "CLIPTextTransformer"):      # transformers 5.x dropped the class transformer_utils.py:5 subclasses for an
        hf_clip.CLIPTextTransformer = type("CLIPTextTransformer", (nn.Module,), {})   # embedder that is not on this path (CLIPTextTransformer_M)
    ref_import.install()
    for name in ("clip", "kornia"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    oc = types.ModuleType("open_clip")
    oc.create_model_and_transforms, oc.tokenize = create_model_and_transforms, None
    sys.modules["open_clip"] = oc
    mods = ref_import.ref("ldm.modules.encoders.modules")
    tokens = torch.zeros(3, 77, dtype=torch.long)
    tokens[0, 0], tokens[0, 1] = 510, 511                                 # the empty prompt: [SOT, EOT, 0...]
    g = torch.Generator().manual_seed(7)
    tokens[1, 0] = 510
    tokens[1, 1:40] = torch.randint(1, 510, (39,), generator=g)
    tokens[1, 40] = 511
    tokens[2, 0] = 510
    tokens[2, 1:76] = torch.randint(1, 510, (75,), generator=g)          # a full-length prompt (EOT in the last slot)
    tokens[2, 76] = 511
    out = {"tokens": tokens}
    for layer in ("last", "penultimate"):
        emb = mods.FrozenOpenCLIPEmbedder(arch="ViT-H-14", version="laion2b_s32b_b79k", device="cpu", layer=layer)
        assert made[-1] == ("ViT-H-14", "laion2b_s32b_b79k") and not hasattr(emb.model, "visual")
        synth.fill_module_(emb, "clip")
        with torch.no_grad():
            emb.model.positional_embedding.mul_(10.0)                    # (the synthetic 1-D recipe is tiny: give the embeddings scale;
            emb.model.token_embedding.weight.mul_(10.0)                  #  tests/test_nets_gpu.py::test_text_tower_vs_oracle does the same)
            out[layer] = emb.encode_with_transformer(tokens)
        out["names_shapes"] = names_shapes(emb)
    save("g_text_openclip", **out)


def gen_signatures():
    """Argument lists of the reference's public entry points on this path (interface data for the drop-in check in
    tests/test_host_cpu.py::test_drop_in_signatures): read with `ast` from the reference sources, nothing is imported."""
    import ast
    want = {
        "ldm/models/diffusion/ddpm.py": {"LatentDiffusionVSRTextWT": ["sample", "sample_canvas", "p_sample_loop", "p_sample_loop_canvas", "compute_flow",
                                                                      "compute_temporal_condition_v4", "apply_model", "get_learned_conditioning",
                                                                      "encode_first_stage", "_gaussian_weights", "p_sample", "p_sample_canvas",
                                                                      "p_mean_variance", "p_mean_variance_canvas", "decode_first_stage"],
                                         "DDPM": ["q_sample", "q_sample_respace", "predict_start_from_noise", "q_posterior", "register_schedule"]},
        "ldm/modules/diffusionmodules/openaimodel.py": {"InflatedUNetModelDualcondV2": ["forward"], "InflatedEncoderUNetModelWT": ["forward"]},
        "ldm/models/autoencoder.py": {"VideoAutoencoderKLResi": ["encode", "decode", "init_from_ckpt"], "AutoencoderKL": ["encode", "decode", "init_from_ckpt"]},
        "ldm/modules/encoders/modules.py": {"FrozenOpenCLIPEmbedder": ["__init__", "freeze", "forward", "encode", "encode_with_transformer"]},
        "scripts/util_image.py": {"ImageSpliterTh": ["__init__", "extract_starts", "update", "gather"]},
        "basicsr/archs/raft_arch.py": {"RAFT_SR": ["__init__", "forward"]},
        "basicsr/archs/arch_util.py": {None: ["flow_warp", "resize_flow"]},
        "scripts/util_flow.py": {None: ["forward_backward_consistency_check"]},
        "scripts/wavelet_color_fix.py": {None: ["adaptive_instance_normalization", "wavelet_reconstruction"]},   # what the scripts import
    }
    out = {}
    for rel, classes in want.items():
        tree = ast.parse(open(os.path.join(ref_import.REF, rel)).read())
        for cname, fns in classes.items():
            body = tree.body if cname is None else next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cname).body
            for f in body:
                if isinstance(f, ast.FunctionDef) and f.name in fns:
                    args = [a.arg for a in f.args.args if a.arg != "self"]
                    out[f"{rel}:{cname or ''}:{f.name}"] = args
    missing = [f"{rel}:{c or ''}:{f}" for rel, cl in want.items() for c, fs in cl.items() for f in fs if f"{rel}:{c or ''}:{f}" not in out]
    assert not missing, missing
    with open(os.path.join(OUT, "g_signatures.json"), "w") as fh:
        json.dump(out, fh, indent=0, sort_keys=True)
    print("wrote g_signatures.json", len(out))


def gen_spliter():
    """ImageSpliterTh of the reference (scripts/util_image.py:686-769): start lists for the script's default patch settings
    and ragged sizes, and — on a small 2-frame image — the iteration order (patches + index tuples), update() accumulation
    and the uniform-count gather(), with sf = 1 and sf = 2."""
    ui = ref_import.ref("scripts.util_image")
    out = {}
    for L, size, stride in [(1024, 960, 750), (512, 960, 750), (2000, 960, 750), (100, 64, 32), (960, 960, 750), (961, 960, 750)]:
        sp = ui.ImageSpliterTh(torch.zeros(1, 1, L, L + 8), size, stride, sf=1)
        out[f"h_{L}_{size}_{stride}"] = np.array(sp.height_starts_list)
        out[f"w_{L}_{size}_{stride}"] = np.array(sp.width_starts_list)
    for sf in (1, 2):
        im = synth.synth_tensor(f"spliter/im{sf}", (2, 3, 40, 52))
        sp = ui.ImageSpliterTh(im, 24, 16, sf=sf)
        idx, k = [], 0
        for pch, index_infos in sp:
            # stand-in for the per-patch model: a smooth function of the patch, upsampled by sf
            res = torch.nn.functional.interpolate(pch * (1.0 + 0.1 * k) + 0.01 * k, scale_factor=sf, mode="nearest")
            sp.update(res, index_infos)
            idx.append(list(index_infos))
            k += 1
        out[f"it_im_sf{sf}"], out[f"it_index_sf{sf}"], out[f"it_gather_sf{sf}"] = im, np.array(idx), sp.gather()
    save("g_spliter", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:            # e.g. `make_golden.py raft`: regenerate selected fixtures only
        for what in sys.argv[1:]:
            if what.startswith("workload:"):          # workload:<case>:<steps>
                gen_workload(*what.split(":")[1:])
                continue
            fn = globals()["gen_" + what]
            if what in ("pstep", "sample_opts"):
                fn(*build_ref_model())
            else:
                fn()
        sys.exit(0)
    gen_raft()
    gen_flow()
    gen_guidance()
    gen_spliter()
    gen_vae()
    model, ddpm = build_ref_model()
    gen_schedule(model)
    gen_unet(model)
    gen_first_stage(model)
    gen_sample(model, ddpm)
    gen_pstep(model, ddpm)
    gen_sample_opts(model, ddpm)
