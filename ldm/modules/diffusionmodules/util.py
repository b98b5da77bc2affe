"""`ldm.modules.diffusionmodules.util` surface (reference util.py:21-43, 96-99, 199-216, 291-310)."""
from mgld_vsr_amd.ddpm import extract_into_tensor, make_beta_schedule  # noqa: F401
from mgld_vsr_amd.unet import GroupNorm32, SpatialTemporalConv, normalization  # noqa: F401
