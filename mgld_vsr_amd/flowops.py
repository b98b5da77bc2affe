"""Flow / colour-fix / tiling free functions of the hot path, on libmgld_hip.

Mirrors basicsr/archs/arch_util.py:156-194 (flow_warp), :235-270 (resize_flow), scripts/util_flow.py:97-136
(flow_warp, forward_backward_consistency_check), scripts/wavelet_color_fix.py:44-119 (AdaIN / wavelet) and
scripts/util_image.py:686-769 (ImageSpliterTh).  Inputs may live on the host or the device; compute is always the
HIP kernels (there is no CPU fallback), results are device tensors.
"""
import torch

from . import hip


def _dev(t):
    hip.lib()
    if not torch.cuda.is_available():
        raise RuntimeError("mgld_vsr_amd needs a HIP device: no CPU fallback on the product path")
    return t.to("cuda", torch.float32).contiguous()


def flow_warp(x, flow, interp_mode="bilinear", padding_mode="zeros", align_corners=True, return_mask=False):
    """arch_util.flow_warp: x [n,c,h,w], flow [n,h,w,2] (dx,dy)."""
    assert interp_mode == "bilinear" and padding_mode == "zeros" and align_corners, "only the configuration the VSR path uses"
    assert x.size()[-2:] == flow.size()[1:3]
    xd = _dev(x)
    fd = _dev(flow.permute(0, 3, 1, 2))
    out = hip.flow_warp(xd, fd, torch.empty_like(xd))
    if not return_mask:
        return out
    mask = hip.flow_warp(torch.ones_like(xd), fd, torch.empty_like(xd))
    mask = (mask >= 0.9999).to(out.dtype)
    return out, mask


def flow_warp_n2hw(feature, flow, mask=False, mode="bilinear", padding_mode="zeros"):
    """scripts/util_flow.flow_warp: flow [n,2,h,w]."""
    assert mode == "bilinear" and padding_mode == "zeros" and not mask
    xd, fd = _dev(feature), _dev(flow)
    return hip.flow_warp(xd, fd, torch.empty_like(xd))


def resize_flow(flow, size_type, sizes, interp_mode="bilinear", align_corners=False):
    assert interp_mode == "bilinear" and not align_corners
    _, _, fh, fw = flow.size()
    if size_type == "ratio":
        oh, ow = int(fh * sizes[0]), int(fw * sizes[1])
    elif size_type == "shape":
        oh, ow = sizes[0], sizes[1]
    else:
        raise ValueError(f"Size type should be ratio or shape, but got type {size_type}.")
    fd = _dev(flow)
    return hip.resize_flow(fd, torch.empty(fd.shape[0], 2, oh, ow, device=fd.device))


def forward_backward_consistency_check(fwd_flow, bwd_flow, alpha=0.01, beta=0.5):
    assert fwd_flow.dim() == 4 and bwd_flow.dim() == 4 and fwd_flow.size(1) == 2 and bwd_flow.size(1) == 2
    f, b = _dev(fwd_flow), _dev(bwd_flow)
    n, _, h, w = f.shape
    fo, bo = torch.empty(n, h, w, device=f.device), torch.empty(n, h, w, device=f.device)
    hip.fb_consistency(f, b, alpha, beta, fo, bo)
    return fo, bo


def adaptive_instance_normalization(content_feat, style_feat):
    c, s = _dev(content_feat), _dev(style_feat)
    assert c.dim() == 4 and c.shape[:2] == s.shape[:2]
    work = torch.empty(512 * c.shape[0] * c.shape[1] + 8, dtype=torch.float32, device=c.device)   # mgld_adain: 512 floats per plane
    if s.shape != c.shape:
        raise NotImplementedError("AdaIN expects content and style of the same shape (as the VSR scripts pass)")
    return hip.adain(c, s, torch.empty_like(c), work)


def wavelet_reconstruction(content_feat, style_feat):
    c, s = _dev(content_feat), _dev(style_feat)
    assert c.shape == s.shape
    work = torch.empty(3 * c.numel(), dtype=torch.float32, device=c.device)
    return hip.wavelet_reconstruction(c, s, torch.empty_like(c), work)


class ImageSpliterTh:
    """scripts/util_image.py:686-769: pixel tiling with uniform-count averaging.  Device-resident fp32 images are cropped,
    accumulated and normalised by the C-ABI kernels (mgld_crop / mgld_tile_accumulate / mgld_tile_normalize); host tensors keep
    the reference's tensor arithmetic (bookkeeping / tests)."""

    def __init__(self, im, pch_size, stride, sf=1):
        assert stride <= pch_size
        self.stride, self.pch_size, self.sf = stride, pch_size, sf
        bs, chn, height, width = im.shape
        self.height_starts_list = self.extract_starts(height)
        self.width_starts_list = self.extract_starts(width)
        self.length = len(self)
        self.num_pchs = 0
        self.im_ori = im
        self.im_res = torch.zeros([bs, chn, height * sf, width * sf], dtype=im.dtype, device=im.device)
        self.pixel_count = torch.zeros([bs, chn, height * sf, width * sf], dtype=im.dtype, device=im.device)

    def extract_starts(self, length):
        """patch origins along one axis: a stride grid whose patches are pulled back inside the image, first occurrence kept
        (pinned by tests/golden/g_spliter.npz, generated from the reference class)"""
        last = max(length - self.pch_size, 0)
        seen = []
        for origin in range(0, length, self.stride):
            origin = min(origin, last)
            if origin not in seen:
                seen.append(origin)
        return seen[:1] if length <= self.pch_size else seen

    def __len__(self):
        return len(self.height_starts_list) * len(self.width_starts_list)

    def __iter__(self):
        return self

    def __next__(self):
        if self.num_pchs >= self.length:
            raise StopIteration()
        w_start = self.width_starts_list[self.num_pchs // len(self.height_starts_list)]
        h_start = self.height_starts_list[self.num_pchs % len(self.height_starts_list)]
        ph, pw = min(self.pch_size, self.im_ori.shape[2]), min(self.pch_size, self.im_ori.shape[3])
        if self._on_device(self.im_ori):
            pch = torch.empty(self.im_ori.shape[0], self.im_ori.shape[1], ph, pw, device=self.im_ori.device)
            hip.crop(self.im_ori.contiguous(), pch, h_start, w_start)
        else:
            pch = self.im_ori[:, :, h_start:h_start + self.pch_size, w_start:w_start + self.pch_size]
        self.h_start, self.h_end = h_start * self.sf, (h_start + self.pch_size) * self.sf
        self.w_start, self.w_end = w_start * self.sf, (w_start + self.pch_size) * self.sf
        self.num_pchs += 1
        return pch, (self.h_start, self.h_end, self.w_start, self.w_end)

    def update(self, pch_res, index_infos):
        if index_infos is None:
            h_start, h_end, w_start, w_end = self.h_start, self.h_end, self.w_start, self.w_end
        else:
            h_start, h_end, w_start, w_end = index_infos
        if self._on_device(self.im_res):
            p = pch_res.to(self.im_res.device, torch.float32).contiguous()
            ones = torch.ones(p.shape[2], p.shape[3], device=p.device)
            hip.tile_accumulate(p, ones, self.im_res, self.pixel_count, h_start, w_start)
            return
        self.im_res[:, :, h_start:h_end, w_start:w_end] += pch_res.to(self.im_res.device)
        self.pixel_count[:, :, h_start:h_end, w_start:w_end] += 1

    def gather(self):
        if self._on_device(self.im_res):
            return hip.tile_normalize(self.im_res, self.pixel_count, torch.empty_like(self.im_res))
        assert torch.all(self.pixel_count != 0)
        return self.im_res.div(self.pixel_count)

    @staticmethod
    def _on_device(t):
        return t.is_cuda and t.dtype == torch.float32
