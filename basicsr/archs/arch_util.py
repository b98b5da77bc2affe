"""`basicsr.archs.arch_util` surface (reference arch_util.py:156-194, 235-270)."""
from mgld_vsr_amd.flowops import flow_warp, resize_flow  # noqa: F401
