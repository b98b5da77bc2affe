"""`ldm.modules.encoders.modules` surface (reference modules.py:140-199)."""
from mgld_vsr_amd.text import FrozenOpenCLIPEmbedder  # noqa: F401
