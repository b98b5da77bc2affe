"""Oracle (test infrastructure): the reverse-diffusion loop of LatentDiffusionVSRTextWT, plain and tiled.

Follows ldm/models/diffusion/ddpm.py:4501-4599 (p_sample_loop), :4325-4380 (p_sample), :4157-4189 (p_mean_variance),
:4619-4693 / :4383-4442 / :4191-4322 (canvas variants), :4601-4616 (_gaussian_weights).  Noise is an explicit input
(the reference draws torch.randn per step, ddpm.py:4344).
"""
import numpy as np
import torch

from . import flow as oflow
from . import nets
from . import schedule as osched


def gaussian_weights(tile_width, tile_height):
    """ddpm.py:4601-4616 (note the asymmetric midpoints: (W-1)/2 for x, H/2 for y); float64."""
    var = 0.01
    mid = (tile_width - 1) / 2
    xs = [np.exp(-(x - mid) * (x - mid) / (tile_width * tile_width) / (2 * var)) / np.sqrt(2 * np.pi * var)
          for x in range(tile_width)]
    mid = tile_height / 2
    ys = [np.exp(-(y - mid) * (y - mid) / (tile_height * tile_height) / (2 * var)) / np.sqrt(2 * np.pi * var)
          for y in range(tile_height)]
    return torch.tensor(np.outer(ys, xs))


def tile_origins(h, w, tile_size, tile_overlap):
    """Tile enumeration of p_mean_variance_canvas (ddpm.py:4205-4236): list of (y0, x0) in the reference's order."""
    def count(L):
        n, cur = 0, 0
        while cur < L:
            cur = max(n * tile_size - tile_overlap * n, 0) + tile_size
            n += 1
        return n
    rows, cols = count(w), count(h)
    out = []
    ofs_x = ofs_y = 0
    for row in range(rows):
        for col in range(cols):
            if col < cols - 1 or row < rows - 1:
                ofs_x = max(row * tile_size - tile_overlap * row, 0)
                ofs_y = max(col * tile_size - tile_overlap * col, 0)
            if row == rows - 1:
                ofs_x = w - tile_size
            if col == cols - 1:
                ofs_y = h - tile_size
            out.append((ofs_y, ofs_x))
    return out


def eps_model(unet_sd, unet_cfg, sc_sd, sc_cfg, x, lat, t_net, ctx):
    tt = torch.full((x.shape[0],), int(t_net), dtype=torch.long)
    sc = nets.structcond_forward(sc_sd, sc_cfg, lat, tt)
    return nets.unet_forward(unet_sd, unet_cfg, x, tt, ctx, sc)


def lr_guidance_flows(raft_sd, lr_images, T, h, w):
    """compute_temporal_condition_v2 (ddpm.py:3469-3500) up to its loss: LR frames [(b t),3,H,W] -> bicubic resize to the latent grid ->
    compute_flow -> (flows, all-valid masks) in temporal_condition_v4's layout — v2's chain is v4's with nothing occluded."""
    import torch.nn.functional as F
    from . import raft as oraft
    res = F.interpolate(lr_images, size=(h, w), mode="bicubic")
    f_f, f_b = oraft.compute_flow(raft_sd, res.reshape(-1, T, *res.shape[1:]))
    z = torch.zeros(f_f.shape[0], T - 1, 1, h, w)
    return (f_f, f_b), (z, z)


def sample(unet_sd, unet_cfg, sc_sd, sc_cfg, ctx, lat, x_T, noises, steps, guidance_scale=-10.0, flows=None, masks=None,
           tile=None, eps_fn=None, return_all=False, lr_images=None, raft_sd=None):
    """x_T -> x_0.  tile=(tile_size, tile_overlap) selects the aggregation-sampling path.  eps_fn overrides the network
    (used to test the sampler arithmetic in isolation).  lr_images (+ raft_sd): the other guidance term (ddpm.py:4359-4366), applied
    before the flows / masks one as the reference does."""
    _, buf, ori = osched.respaced_schedule(steps)
    T = unet_cfg["num_frames"]
    lr_flows = lr_masks = None
    if lr_images is not None:
        with torch.no_grad():
            lr_flows, lr_masks = lr_guidance_flows(raft_sd, lr_images, T, x_T.shape[2], x_T.shape[3])
    img = x_T.clone()
    traj = []
    with torch.no_grad():
        for k, i in enumerate(reversed(range(steps))):
            t_net = ori[i]
            f = eps_fn or (lambda xx, ll: eps_model(unet_sd, unet_cfg, sc_sd, sc_cfg, xx, ll, t_net, ctx))
            if tile is None:
                eps = f(img, lat)
            else:
                ts, ov = tile
                wgt = gaussian_weights(ts, ts).to(torch.float64)
                acc = torch.zeros_like(img)
                cnt = torch.zeros_like(img)
                for (y0, x0) in tile_origins(img.shape[2], img.shape[3], ts, ov):
                    e = f(img[:, :, y0:y0 + ts, x0:x0 + ts], lat[:, :, y0:y0 + ts, x0:x0 + ts])
                    # float32 canvas += float32 tile * float64 weights (ddpm.py:4295-4296)
                    acc[:, :, y0:y0 + ts, x0:x0 + ts] += e * wgt
                    cnt[:, :, y0:y0 + ts, x0:x0 + ts] += wgt
                eps = acc / cnt
            z, logvar = osched.p_step(buf, i, img, eps, noises[k])
            if lr_flows is not None:
                z, _ = oflow.guidance_update(z, lr_flows, lr_masks, T, guidance_scale, logvar)
            if flows is not None:
                z, _ = oflow.guidance_update(z, flows, masks, T, guidance_scale, logvar)
            img = z
            traj.append(img)
    return (img, traj) if return_all else img
