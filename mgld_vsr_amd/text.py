"""Text-conditioning boundary (reference ldm/modules/encoders/modules.py:140-199, FrozenOpenCLIPEmbedder).

The OpenCLIP ViT-H tower is outside the hot path: the VSR scripts call it once per segment
with the empty prompt, so its output is a CONSTANT [n,77,1024] tensor.  This class keeps the import path / call
signature; without open_clip weights it returns the deterministic synthetic context every BASELINE config uses."""
import warnings

import torch
import torch.nn as nn

from . import synth


class FrozenOpenCLIPEmbedder(nn.Module):
    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", max_length=77, freeze=True,
                 layer="last", context_dim=1024):
        super().__init__()
        self.device, self.max_length, self.layer, self.context_dim = device, max_length, layer, context_dim
        self._context = None

    def set_context(self, ctx):
        """install a precomputed empty-prompt embedding [1,77,context_dim]"""
        self._context = ctx

    def forward(self, text):
        n = len(text) if isinstance(text, (list, tuple)) else 1
        if self._context is None:
            warnings.warn("FrozenOpenCLIPEmbedder: no OpenCLIP weights on this path; using the synthetic constant context")
            self._context = synth.synth_tensor("ctx", (1, self.max_length, self.context_dim))
        return self._context.repeat(n, 1, 1)

    def encode(self, text):
        return self(text)
