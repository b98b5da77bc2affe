"""`scripts.wavelet_color_fix` surface (reference wavelet_color_fix.py:44-119)."""
from mgld_vsr_amd.flowops import adaptive_instance_normalization, wavelet_reconstruction  # noqa: F401
