"""`ldm.models.autoencoder` surface (reference autoencoder.py:299, 1564-1690) -> mgld_vsr_amd.vae."""
from mgld_vsr_amd.vae import AutoencoderKL, VideoAutoencoderKLResi  # noqa: F401
