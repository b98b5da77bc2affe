"""Inputs of the full-width workload cases (BASELINE configs[1]-[4]) shared by the golden generator (make_golden.py, build
container, runs the reference) and the tests (GPU box, runs the product): everything is regenerated from the synth
recipes, so the fixtures hold expected outputs only.

    name      workload                                                                        BASELINE config
    c2        8 frames 512x512, S steps, flows off (model.sample)                             configs[1]; configs[4] (video decoder + AdaIN at T = 8)
    c2g       8 frames 512x512, S steps, flow-guided (guidance_scale -10)                     configs[2] (the per-GPU share of the 32-frame clip)
    c4        4 frames 1024x1024, S steps, flow-guided aggregation sampling                   configs[3] (sample_canvas, tile 64 / overlap 32: nine tiles)
    c2s       8 SMOOTH frames 512x512 translating by (1, 2) latent pixels per frame, guided    configs[2] on realistic content: low-pass LR frames,
              with the flows of that translation (occlusion masks valid except at the border)  bicubic x4 as the scripts do; the guidance term is ACTIVE
"""
import torch

from mgld_vsr_amd import synth

CASES = {
    "c2": dict(T=8, H=512, guided=False, canvas=False, stride=8),
    "c2g": dict(T=8, H=512, guided=True, canvas=False, stride=8),
    "c4": dict(T=4, H=1024, guided=True, canvas=True, stride=16),
    "c2s": dict(T=8, H=512, guided=True, canvas=False, stride=8, smooth=True),
}


def smooth_frames(tag, Tn, H, step=(2, 4)):
    """Tn low-pass, slowly translating frames [Tn,3,H,H] in [-1,1]: a 5x5-box-filtered LR texture (H/4 per side, the recipe of
    make_golden._harness_frames) moving by `step` = (dx, dy) LR pixels per frame, bicubic x4 as the scripts pre-upsample their input."""
    import torch.nn.functional as F
    h = H // 4
    mx, my = step[0] * (Tn - 1), step[1] * (Tn - 1)
    base = F.avg_pool2d(torch.sigmoid(synth.synth_tensor(f"{tag}/xs", (1, 3, h + my + 8, h + mx + 8), 1.8)), 5, 1, 2)
    lr = torch.stack([base[0, :, 4 + my - step[1] * k:4 + my - step[1] * k + h, 4 + mx - step[0] * k:4 + mx - step[0] * k + h]
                      for k in range(Tn)])
    up = F.interpolate(lr, scale_factor=4, mode="bicubic", align_corners=False)
    return (2.0 * up - 1.0).clamp(-1, 1).contiguous()


def case_inputs(name, S):
    """-> dict(T, S, H, h, x [T,3,H,H], noise {posterior, x_T, steps[S]}, ff, fb (latent-resolution flows or None))"""
    c = CASES[name]
    Tn, H = c["T"], c["H"]
    h = H // 8
    tag = name[:2]                                   # c2, c2g and c2s share the noise; c2 / c2g also the frames
    if c.get("smooth"):
        x = smooth_frames(tag, Tn, H)
    else:
        x = synth.synth_tensor(f"{tag}/x", (Tn, 3, H, H), 0.5).clamp(-1, 1)
    noise = {"posterior": synth.synth_tensor(f"{tag}/np", (Tn, 4, h, h)), "x_T": synth.synth_tensor(f"{tag}/n0", (Tn, 4, h, h)),
             "steps": torch.stack([synth.synth_tensor(f"{tag}/n{i}", (Tn, 4, h, h)) for i in range(S)])}
    ff = fb = None
    if c.get("smooth"):                              # content moves by +(1, 2) latent pixels per frame: consistent flow pair
        d = torch.tensor([1.0, 2.0]).view(1, 2, 1, 1)
        ff, fb = d.expand(Tn - 1, 2, h, h).contiguous(), (-d).expand(Tn - 1, 2, h, h).contiguous()
    elif c["guided"]:
        ff, fb = synth.smooth_flow(f"{tag}/ff", Tn - 1, h, h), synth.smooth_flow(f"{tag}/fb", Tn - 1, h, h)
    return dict(T=Tn, S=S, H=H, h=h, x=x, noise=noise, ff=ff, fb=fb, canvas=c["canvas"], stride=c["stride"])


def sample_lr_inputs(T, S=4, h=128, w=128):
    """inputs of the `lr_images` guidance fixture (make_golden.py::gen_sample_lr / g_sample_lr.npz holds the reference's OUTPUTS only):
    struct-cond latent, x_T, per-step noise (loop order), T smooth LR frames 2h x 2w in [0,1] translating by (2, 3) pixels per frame, a
    second (smooth synthetic) flow pair for the combined run"""
    import torch.nn.functional as F
    lat = synth.synth_tensor("samplelr/lat", (T, 4, h, w), 0.5)
    xT = synth.synth_tensor("samplelr/xT", (T, 4, h, w))
    noises = [synth.synth_tensor(f"samplelr/noise{i}", (T, 4, h, w)) for i in range(S)]
    H, W = 2 * h, 2 * w
    base = F.avg_pool2d(torch.sigmoid(synth.synth_tensor("samplelr/lr", (1, 3, H + 16, W + 16), 1.8)), 5, 1, 2)
    lr = torch.stack([base[0, :, 4 + 3 * k:4 + 3 * k + H, 6 + 2 * k:6 + 2 * k + W] for k in range(T)]).contiguous()
    ff, fb = synth.smooth_flow("samplelr/ff", T - 1, h, w, 0.4), synth.smooth_flow("samplelr/fb", T - 1, h, w, 0.4)
    return dict(lat=lat, xT=xT, noises=noises, lr=lr, ff=ff, fb=fb, h=h, w=w, S=S)
