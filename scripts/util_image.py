"""`scripts.util_image` surface (reference util_image.py:686-769)."""
from mgld_vsr_amd.flowops import ImageSpliterTh  # noqa: F401
