"""`ldm.modules.diffusionmodules.openaimodel` surface (reference openaimodel.py:133-594, 1903-2525)."""
from mgld_vsr_amd.unet import (AttentionBlock, Downsample, InflatedEncoderUNetModelWT, InflatedUNetModelDualcondV2,  # noqa: F401
                               QKVAttentionLegacy, ResBlock, ResBlockDual, TimestepEmbedSequential, Upsample)
