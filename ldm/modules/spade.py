"""`ldm.modules.spade` surface (reference spade.py:68-111)."""
from mgld_vsr_amd.unet import SPADE  # noqa: F401
