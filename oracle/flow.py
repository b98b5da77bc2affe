"""Oracle (test infrastructure): flow warp, forward/backward consistency, flow resize, motion-guidance loss.

Follows basicsr/archs/arch_util.py:156-194 (flow_warp), :235-270 (resize_flow), scripts/util_flow.py:45-136
(coords_grid / bilinear_sample / flow_warp / forward_backward_consistency_check) and
ldm/models/diffusion/ddpm.py:3538-3574 (compute_temporal_condition_v4).
"""
import torch
import torch.nn.functional as F


def flow_warp(x, flow_nhw2):
    """x [n,c,h,w], flow [n,h,w,2] (dx,dy) -> bilinear, zeros padding, align_corners=True (arch_util.py:156-184)."""
    _, _, h, w = x.shape
    gy, gx = torch.meshgrid(torch.arange(0, h).type_as(x), torch.arange(0, w).type_as(x), indexing="ij")
    grid = torch.stack((gx, gy), 2).float()
    vgrid = grid + flow_nhw2
    vx = 2.0 * vgrid[:, :, :, 0] / max(w - 1, 1) - 1.0
    vy = 2.0 * vgrid[:, :, :, 1] / max(h - 1, 1) - 1.0
    return F.grid_sample(x, torch.stack((vx, vy), dim=3), mode="bilinear", padding_mode="zeros", align_corners=True)


def flow_warp_n2hw(feature, flow):
    """util_flow.py:97-111 variant: flow [n,2,h,w]."""
    b, c, h, w = feature.shape
    y, x = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    grid = torch.stack([x, y], 0).float()[None].repeat(b, 1, 1, 1) + flow
    xg = 2 * grid[:, 0] / (w - 1) - 1
    yg = 2 * grid[:, 1] / (h - 1) - 1
    return F.grid_sample(feature, torch.stack([xg, yg], -1), mode="bilinear", padding_mode="zeros", align_corners=True)


def forward_backward_consistency_check(fwd_flow, bwd_flow, alpha=0.01, beta=0.5):
    """util_flow.py:114-136."""
    mag = torch.norm(fwd_flow, dim=1) + torch.norm(bwd_flow, dim=1)
    warped_bwd = flow_warp_n2hw(bwd_flow, fwd_flow)
    warped_fwd = flow_warp_n2hw(fwd_flow, bwd_flow)
    diff_fwd = torch.norm(fwd_flow + warped_bwd, dim=1)
    diff_bwd = torch.norm(bwd_flow + warped_fwd, dim=1)
    thr = alpha * mag + beta
    return (diff_fwd > thr).float(), (diff_bwd > thr).float()


def resize_flow(flow, out_h, out_w):
    """arch_util.py:235-270 with size_type='shape', bilinear, align_corners=False."""
    _, _, h, w = flow.shape
    f = flow.clone()
    f[:, 0] *= out_w / w
    f[:, 1] *= out_h / h
    return F.interpolate(f, size=(out_h, out_w), mode="bilinear", align_corners=False)


def temporal_condition_v4(flows, latents, masks, num_frames):
    """ddpm.py:3538-3574. flows = (flow_fwd_prop, flow_bwd_prop) each [b,t-1,2,h,w]; masks = (fwd_occs, bwd_occs)
    each [b,t-1,1,h,w]; latents [(b t),c,h,w]. Note the zero reference of the first comparison of each chain."""
    flow_fwd_prop, flow_bwd_prop = flows
    fwd_occs, bwd_occs = masks
    t = num_frames
    bt, c, h, w = latents.shape
    lat = latents.reshape(bt // t, t, c, h, w)
    loss_b = 0
    cur_warp = torch.zeros_like(lat[:, -1])
    prev = None
    for i in range(t - 1, -1, -1):
        cur = lat[:, i]
        if i < t - 1:
            cur_warp = flow_warp(cur, flow_bwd_prop[:, i].permute(0, 2, 3, 1))
            loss_b = loss_b + F.l1_loss((1 - fwd_occs[:, i]) * prev, (1 - fwd_occs[:, i]) * cur)
        prev = cur_warp
    loss_f = 0
    cur_warp = torch.zeros_like(lat[:, 0])
    for i in range(0, t):
        cur = lat[:, i]
        if i > 0:
            cur_warp = flow_warp(cur, flow_fwd_prop[:, i - 1].permute(0, 2, 3, 1))
            loss_f = loss_f + F.l1_loss((1 - bwd_occs[:, i - 1]) * prev, (1 - bwd_occs[:, i - 1]) * cur)
        prev = cur_warp
    return loss_b + loss_f


def guidance_update(latents, flows, masks, num_frames, guidance_scale, model_log_variance):
    """ddpm.py:4367-4373: latents - guidance_scale * logvar * d(loss)/d(latents)."""
    with torch.enable_grad():
        z = latents.detach().clone().requires_grad_(True)
        loss = temporal_condition_v4(flows, z, masks, num_frames)
        grad = torch.autograd.grad(loss, z)[0]
    return (latents - guidance_scale * model_log_variance * grad).detach(), float(loss.detach())
