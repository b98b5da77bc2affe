#!/usr/bin/env python
"""GPU probe: BASELINE configs[0] (one 512x512 frame, 4 steps, full width) against the reference fixture g_full_c1.npz under several
MGLD_W2 precision-scope settings: latent / frame / decoder-only rel-L2 and the decode time.
    python tools/prec_probe_c1.py "setting;setting;..."    -> gpurun_out/prec_probe_c1.json"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def rel_l2(a, b):
    a, b = a.detach().cpu().double().flatten(), b.detach().cpu().double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def main():
    from test_oracle_golden import fullwidth_c1_inputs
    from mgld_vsr_amd.pipeline import VSRPipeline, model_configs
    settings = sys.argv[1].split(";")
    d = np.load(os.path.join(ROOT, "tests", "golden", "g_full_c1.npz"))
    g = {k: torch.from_numpy(d[k]) for k in d.files if d[k].dtype.kind in "fiu"}
    Tn, S, H, h, x, noise = fullwidth_c1_inputs()
    res = {}
    for setting in settings:
        os.environ["MGLD_W2"] = setting
        pipe = VSRPipeline(num_frames=Tn, ddpm_steps=S, configs=model_configs(Tn))
        out, lat = pipe.run_segment(x, noise=noise, return_latents=True)
        vq = pipe.vq_model
        _, fea = vq.encode(x.cuda())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        vq.decode(g["x0"].cuda() / 0.18215, fea)
        e0.record()
        dec = vq.decode(g["x0"].cuda() / 0.18215, fea)
        e1.record()
        torch.cuda.synchronize()
        r = {"latent": rel_l2(lat, g["x0"]), "frames": rel_l2(out[:, :, ::4, ::4], g["out_s4"]), "decoder_only": rel_l2(dec[:, :, ::4, ::4], g["dec_s4"]),
             "decode_ms_T1": e0.elapsed_time(e1)}
        res[setting] = r
        print(setting, json.dumps(r), flush=True)
        del pipe, vq, out, lat, dec, fea
        torch.cuda.empty_cache()
    od = os.path.join(ROOT, "gpurun_out")
    os.makedirs(od, exist_ok=True)
    with open(os.path.join(od, "prec_probe_c1.json"), "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
