"""Host pre/post-processing of the inference script, on the device (kernels K11).

The reference does these steps with torch ops on host tensors around the hot path
(scripts/vsr_val_ddpm_text_T_vqganfin_oldcanvas_tile.py:349-357 bicubic pre-upsampling, :384-390 reflect padding to a
multiple of 32, :393-396 the [0,1] quarter-resolution copy fed to the flow network, :523-543 crop + uint8 conversion).
Here they are C-ABI launches (mgld_resize_bicubic / mgld_reflect_pad / mgld_to_uint8_hwc) on device-resident frames, so a
segment crosses PCIe once as uint8-sized LR frames and once as the uint8 result.
"""
import torch

from . import hip


def upsample_lr(frames, upscale=4.0, device="cuda"):
    """frames: [T,3,h,w] in [-1,1] (read_image output).  Bicubic pre-upsampling by max(512/min(h,w), upscale) as the script
    does per frame (:349-357), clamped to [-1,1] (:376)."""
    x = frames.to(device, torch.float32)
    h, w = x.shape[-2:]
    s = max(512.0 / min(h, w), float(upscale))
    return hip.resize_bicubic(x, (int(h * s), int(w * s)), clamp=(-1.0, 1.0))


def pad_to_32(x):
    """reflect-pad bottom/right to a multiple of 32 (:381-390).  As in the script, as soon as ONE side is not a multiple of
    32 both get `(side // 32 + 1) * 32 - side` (a side that already is a multiple grows by 32).  Returns (padded, ori_h, ori_w)."""
    h, w = x.shape[-2:]
    if h % 32 == 0 and w % 32 == 0:
        return x, h, w
    return hip.reflect_pad(x, (h // 32 + 1) * 32, (w // 32 + 1) * 32), h, w


def flow_input(x):
    """[0,1] quarter-resolution frames for the flow network (:392-396)."""
    h, w = x.shape[-2:]
    x01 = torch.clamp((x + 1.0) / 2.0, min=0.0, max=1.0)
    return hip.resize_bicubic(x01, (h // 4, w // 4))


def to_png_payload(out, ori_h, ori_w):
    """out [T,3,H,W] in [0,1] on the device -> uint8 [T,ori_h,ori_w,3] on the host (crop of the padding + rounding)."""
    return hip.to_uint8_hwc(out, ori_h, ori_w).cpu().numpy()


class FrameWriter:
    """PNG / .npy output off the sampling thread (host I/O overlaps with sampling).  PNG encoding of eight
    512x512 frames costs the host ~0.3 s, a third of a segment's GPU time; zlib releases the GIL, so a small thread pool hides it
    behind the next segment's launches.  `close()` (or leaving the `with` block) waits for every file and re-raises the first error."""

    def __init__(self, workers=4):
        from concurrent.futures import ThreadPoolExecutor
        self._pool = ThreadPoolExecutor(max_workers=max(1, int(workers)))
        self._pending = []
        self._by_path = {}      # a path written twice (the repeat-last padding of a short final segment): the later write wins

    @staticmethod
    def _png(path, arr):
        from PIL import Image
        Image.fromarray(arr).save(path)

    @staticmethod
    def _npy(path, arr):
        import numpy as np
        with open(path, "wb") as fh:
            np.save(fh, arr)

    def _submit(self, fn, path, arr):
        prev = self._by_path.get(path)
        if prev is not None:
            try:
                prev.result()       # never two writers on one file; the earlier error (if any) resurfaces in close()
            except Exception:
                pass
        f = self._pool.submit(fn, path, arr)
        self._by_path[path] = f
        self._pending.append(f)

    def png(self, path, arr):
        self._submit(self._png, path, arr)

    def npy(self, path, arr):
        self._submit(self._npy, path, arr)

    def close(self):
        err = None
        for f in self._pending:
            try:
                f.result()
            except Exception as e:          # keep draining: every file that can be written is written
                err = err or e
        self._pending = []
        self._pool.shutdown(wait=True)
        if err is not None:
            raise err

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
