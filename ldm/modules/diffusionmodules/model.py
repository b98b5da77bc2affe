"""`ldm.modules.diffusionmodules.model` surface (reference model.py:84-244, 473-572, 926-1056, 1312-1367)."""
from mgld_vsr_amd.vae import (AttnBlock, Downsample, Encoder, Fuse_sft_block_ResidualDenseBlock, Normalize, ResBlock,  # noqa: F401
                              ResnetBlock, Upsample, VideoDecoder_Mix, make_attn)
