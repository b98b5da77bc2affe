"""Inputs of the full-width workload cases (BASELINE configs[1]-[4]) shared by the golden generator (make_golden.py, build
container, runs the reference) and the tests (GPU box, runs the product): everything is regenerated from the synth
recipes, so the fixtures hold expected outputs only.

    name      workload                                                                        BASELINE config
    c2        8 frames 512x512, S steps, flows off (model.sample)                             configs[1]; configs[4] (video decoder + AdaIN at T = 8)
    c2g       8 frames 512x512, S steps, flow-guided (guidance_scale -10)                     configs[2] (the per-GPU share of the 32-frame clip)
    c4        4 frames 1024x1024, S steps, flow-guided aggregation sampling                   configs[3] (sample_canvas, tile 64 / overlap 32: nine tiles)
"""
import torch

from mgld_vsr_amd import synth

CASES = {
    "c2": dict(T=8, H=512, guided=False, canvas=False, stride=8),
    "c2g": dict(T=8, H=512, guided=True, canvas=False, stride=8),
    "c4": dict(T=4, H=1024, guided=True, canvas=True, stride=16),
}


def case_inputs(name, S):
    """-> dict(T, S, H, h, x [T,3,H,H], noise {posterior, x_T, steps[S]}, ff, fb (latent-resolution flows or None))"""
    c = CASES[name]
    Tn, H = c["T"], c["H"]
    h = H // 8
    tag = name[:2]                                   # c2 and c2g share frames and noise: guidance is the only difference
    x = synth.synth_tensor(f"{tag}/x", (Tn, 3, H, H), 0.5).clamp(-1, 1)
    noise = {"posterior": synth.synth_tensor(f"{tag}/np", (Tn, 4, h, h)), "x_T": synth.synth_tensor(f"{tag}/n0", (Tn, 4, h, h)),
             "steps": torch.stack([synth.synth_tensor(f"{tag}/n{i}", (Tn, 4, h, h)) for i in range(S)])}
    ff = fb = None
    if c["guided"]:
        ff, fb = synth.smooth_flow(f"{tag}/ff", Tn - 1, h, h), synth.smooth_flow(f"{tag}/fb", Tn - 1, h, h)
    return dict(T=Tn, S=S, H=H, h=h, x=x, noise=noise, ff=ff, fb=fb, canvas=c["canvas"], stride=c["stride"])
