"""TEST INFRASTRUCTURE — CPU restatement of the inference script's host pre/post-processing (the reference performs these
steps with plain torch ops; restated here with the same calls, each citing the script line it follows).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product path never does."""
import numpy as np
import torch
import torch.nn.functional as F


def upsample_lr(frames, upscale=4.0):
    """scripts/vsr_val_ddpm_text_T_vqganfin_oldcanvas_tile.py:349-357 (+ the clamp of :376)"""
    h, w = frames.shape[-2:]
    s = max(512 / min(h, w), upscale)
    out = torch.cat([F.interpolate(f[None], size=(int(h * s), int(w * s)), mode="bicubic") for f in frames], 0)
    return out.clamp(-1.0, 1.0)


def pad_to_32(x):
    """:381-390"""
    ori_h, ori_w = x.shape[2:]
    if not (ori_h % 32 == 0 and ori_w % 32 == 0):
        pad_h = ((ori_h // 32) + 1) * 32 - ori_h
        pad_w = ((ori_w // 32) + 1) * 32 - ori_w
        x = F.pad(x, pad=(0, pad_w, 0, pad_h), mode="reflect")
    return x, ori_h, ori_w


def flow_input(x):
    """:392-396"""
    x01 = torch.clamp((x + 1.0) / 2.0, min=0.0, max=1.0)
    _, _, h, w = x01.shape
    return F.interpolate(x01, size=(h // 4, w // 4), mode="bicubic")


def to_png_payload(out, ori_h, ori_w):
    """:523-543 — crop the padding, [0,1] -> uint8 HWC"""
    o = out[:, :, :ori_h, :ori_w]
    return (o.cpu().numpy().transpose(0, 2, 3, 1) * 255).astype(np.uint8)


def resize_center_crop(x, size):
    """torchvision.transforms.Resize(size) + CenterCrop(size) on a tensor as scripts/vsr_val_ddpm_text_T_vqganfin_old.py:253-256,315
    apply them (torchvision 0.13 / 0.14 tensor path: bilinear, align_corners=False, no antialias; smaller edge -> size)."""
    h, w = x.shape[-2:]
    if w <= h:
        nw, nh = size, int(size * h / w)
    else:
        nh, nw = size, int(size * w / h)
    y = F.interpolate(x, size=[nh, nw], mode="bilinear", align_corners=False, antialias=False)
    top, left = int(round((nh - size) / 2.0)), int(round((nw - size) / 2.0))
    return y[..., top:top + size, left:left + size]
