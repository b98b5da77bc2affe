"""`scripts.util_flow` surface (reference util_flow.py:97-136)."""
from mgld_vsr_amd.flowops import flow_warp_n2hw as flow_warp, forward_backward_consistency_check  # noqa: F401
