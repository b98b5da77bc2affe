"""`ldm.modules.distributions.distributions` surface (reference distributions.py:24-40)."""
from mgld_vsr_amd.vae import DiagonalGaussianDistribution  # noqa: F401
