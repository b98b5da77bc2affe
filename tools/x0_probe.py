#!/usr/bin/env python
"""GPU probe: where the error of the sampled latents comes from on a full-width workload fixture.  Runs the segment (a) as shipped,
(b) with the REFERENCE's struct-cond latent / x_T handed to the sampler (fixture `init`, `xT`): the sampler alone, (c) = (b) with the
weight-residual pass on every contraction.  Reports rel-L2 of the first-stage latent and of x_0 for each.
    python tools/x0_probe.py [case] [S] [w2 settings ;-separated]     -> gpurun_out/x0_probe_<case>_S<S>.json"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from cases import case_inputs  # noqa: E402


def rel_l2(a, b):
    a, b = a.detach().cpu().double().flatten(), b.detach().cpu().double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else "c2s"
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    settings = (sys.argv[3] if len(sys.argv) > 3 else "default;all").split(";")
    d = np.load(os.path.join(ROOT, "tests", "golden", f"g_work_{case}_S{S}.npz"))
    g = {k: torch.from_numpy(d[k]) for k in d.files}
    c = case_inputs(case, S)
    Tn = c["T"]
    from mgld_vsr_amd.pipeline import VSRPipeline, model_configs
    res = {}
    for setting in settings:
        if setting == "default":
            os.environ.pop("MGLD_W2", None)
        else:
            os.environ["MGLD_W2"] = setting
        pipe = VSRPipeline(num_frames=Tn, ddpm_steps=S, configs=model_configs(Tn))
        flows = masks = None
        if c["ff"] is not None:
            flows, masks = (c["ff"][None], c["fb"][None]), (g["focc"][None, :, None], g["bocc"][None, :, None])
        x = c["x"].cuda()
        out, lat = pipe.run_segment(x, flows=flows, masks=masks, guidance_scale=-10.0, noise=c["noise"], return_latents=True,
                                    tile=(64, 32) if c["canvas"] else None)
        r = {"init_own": rel_l2(pipe.last_init_latent, g["init"]), "x0_own_init": rel_l2(lat, g["x0"])}
        m = pipe.model
        ctx = m.cond_stage_model([""])
        kw = dict(cond=ctx, struct_cond=g["init"].cuda(), guidance_scale=-10.0, flows=flows, masks=masks, batch_size=1, timesteps=S,
                  time_replace=S, x_T=g["xT"].cuda(), noise=c["noise"]["steps"], use_graph=True)
        x0 = m.sample_canvas(tile_size=64, tile_overlap=32, batch_size_sample=1, **kw) if c["canvas"] else m.sample(**kw)
        r["x0_ref_init"] = rel_l2(x0, g["x0"])
        if flows is not None:            # the same without guidance on BOTH... not available (the fixture is guided); own-vs-own sensitivity instead:
            kw2 = dict(kw, struct_cond=pipe.last_init_latent)
            x0b = m.sample(**kw2) if not c["canvas"] else None
            if x0b is not None:
                r["x0_own_init_ref_xT"] = rel_l2(x0b, g["x0"])
        res[setting] = r
        print(setting, json.dumps(r), flush=True)
        del pipe, m, out, lat, x0
        torch.cuda.empty_cache()
    od = os.path.join(ROOT, "gpurun_out")
    os.makedirs(od, exist_ok=True)
    with open(os.path.join(od, f"x0_probe_{case}_S{S}.json"), "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
