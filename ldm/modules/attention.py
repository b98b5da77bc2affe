"""`ldm.modules.attention` surface (reference attention.py:48-75, 124-143, 262-435, 484-546)."""
from mgld_vsr_amd.unet import (BasicTransformerBlockV2, FeedForward, GEGLU, MemoryEfficientCrossAttention,  # noqa: F401
                               MemoryEfficientSelfAttention, SpatialTransformerV2, TemporalAttention)
