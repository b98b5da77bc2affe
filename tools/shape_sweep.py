"""Shape sweep of the whole per-segment path on reduced-width nets (1..16 frames, non-square, tiled): crash / NaN check."""
import sys, torch
sys.path.insert(0, ".")
from mgld_vsr_amd.pipeline import VSRPipeline, model_configs
small = dict(unet_overrides=dict(model_channels=64, context_dim=64, semb_channels=64), struct_overrides=dict(model_channels=64, out_channels=64, num_heads=1), vae_overrides=dict(ch=32), context_dim=64)
for T, H, W, guided, tile in [(1, 256, 256, False, None), (2, 256, 320, True, None), (3, 256, 256, True, None), (5, 256, 256, True, None), (3, 576, 640, True, (64, 32)), (16, 256, 256, True, None)]:
    pipe = VSRPipeline(num_frames=T, ddpm_steps=3, configs=model_configs(T, **small))
    x = (torch.rand(T, 3, H, W) * 2 - 1).cuda()
    fl = mk = None
    if guided and T > 1:
        fl, mk = pipe.estimate_flows(x)
    out = pipe.run_segment(x, flows=fl, masks=mk, tile=tile)
    print(T, H, W, guided, tile, tuple(out.shape), bool(torch.isfinite(out).all()), float(out.mean()))
