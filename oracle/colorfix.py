"""Oracle (test infrastructure): AdaIN / wavelet colour fix, following scripts/wavelet_color_fix.py:44-119."""
import torch
import torch.nn.functional as F


def calc_mean_std(feat, eps=1e-5):
    b, c = feat.shape[:2]
    var = feat.reshape(b, c, -1).var(dim=2) + eps
    return feat.reshape(b, c, -1).mean(dim=2).reshape(b, c, 1, 1), var.sqrt().reshape(b, c, 1, 1)


def adaptive_instance_normalization(content, style):
    sm, ss = calc_mean_std(style)
    cm, cs = calc_mean_std(content)
    return (content - cm) / cs * ss + sm


def wavelet_blur(image, radius):
    c = image.shape[1]
    k = torch.tensor([[0.0625, 0.125, 0.0625], [0.125, 0.25, 0.125], [0.0625, 0.125, 0.0625]], dtype=image.dtype)
    k = k[None, None].repeat(c, 1, 1, 1)
    image = F.pad(image, (radius, radius, radius, radius), mode="replicate")
    return F.conv2d(image, k, groups=c, dilation=radius)


def wavelet_decomposition(image, levels=5):
    high = torch.zeros_like(image)
    low = image
    for i in range(levels):
        low = wavelet_blur(image, 2 ** i)
        high = high + (image - low)
        image = low
    return high, low


def wavelet_reconstruction(content, style):
    ch, _ = wavelet_decomposition(content)
    _, sl = wavelet_decomposition(style)
    return ch + sl
