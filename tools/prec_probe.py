#!/usr/bin/env python
"""GPU probe: what each precision scope of the weight-residual pass (MGLD_W2, engine.w2_scopes) buys and costs on a full-width workload
fixture.  For every setting: latent / frame / decoder-only rel-L2 against the reference fixture + hipEvent time of encode x2, sample, decode.
    python tools/prec_probe.py [case] [S] [setting;setting;...]      -> gpurun_out/prec_probe_<case>_S<S>.json"""
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
    case = sys.argv[1] if len(sys.argv) > 1 else "c2"
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    settings = (sys.argv[3] if len(sys.argv) > 3 else "0;vae_dec;vae_dec,vae_enc;vae_dec,vae_enc,first,unet_io").split(";")
    d = np.load(os.path.join(ROOT, "tests", "golden", f"g_work_{case}_S{S}.npz"))
    g = {k: torch.from_numpy(d[k]) for k in d.files}
    c = case_inputs(case, S)
    Tn, st = c["T"], c["stride"]
    from mgld_vsr_amd.flowops import adaptive_instance_normalization
    from mgld_vsr_amd.pipeline import VSRPipeline, model_configs
    res = {}
    for setting in settings:
        os.environ["MGLD_W2"] = setting
        pipe = VSRPipeline(num_frames=Tn, ddpm_steps=S, configs=model_configs(Tn))
        flows = masks = None
        if c["ff"] is not None:
            flows, masks = (c["ff"][None], c["fb"][None]), (g["focc"][None, :, None], g["bocc"][None, :, None])
        kw = dict(flows=flows, masks=masks, guidance_scale=-10.0, noise=c["noise"], return_latents=True, tile=(64, 32) if c["canvas"] else None)
        x = c["x"].cuda()
        out, lat = pipe.run_segment(x, **kw)       # warm (caches, graph)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        vq, m = pipe.vq_model, pipe.model
        ev[0].record()
        out, lat = pipe.run_segment(x, **kw)
        ev[1].record()
        _, fea = vq.encode(x)
        ev[2].record()
        dec = vq.decode(g["x0"].cuda() * (1.0 / m.scale_factor), fea)
        ev[3].record()
        vq.decoder.fusion_w = 0.5
        dec05 = vq.decode(lat * (1.0 / m.scale_factor), fea)
        vq.decoder.fusion_w = 1.0
        out05 = torch.clamp((adaptive_instance_normalization(dec05, x) + 1.0) / 2.0, 0.0, 1.0)
        torch.cuda.synchronize()
        r = {"latent": rel_l2(lat, g["x0"]), "frames": rel_l2(out[:, :, ::st, ::st], g["out_s"]),
             "frames_w05": rel_l2(out05[:, :, ::st, ::st], g["out_w05_s"]), "decoder_only": rel_l2(dec[:, :, ::st, ::st], g["dec_s"]),
             "segment_ms": ev[0].elapsed_time(ev[1]), "encode_ms": ev[1].elapsed_time(ev[2]), "decode_ms": ev[2].elapsed_time(ev[3]),
             "launches_per_step": getattr(m, "last_launches_per_step", None), "gn_stats_saved": m.engine().gn_stats_saved}
        res[setting] = r
        print(setting, json.dumps(r), flush=True)
        del pipe, vq, m, out, lat, dec, dec05, out05, fea
        torch.cuda.empty_cache()
    od = os.path.join(ROOT, "gpurun_out")
    os.makedirs(od, exist_ok=True)
    with open(os.path.join(od, f"prec_probe_{case}_S{S}.json"), "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
