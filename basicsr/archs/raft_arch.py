"""`basicsr.archs.raft_arch.RAFT_SR` (reference raft_arch.py:668-807): the shipped YAML's `flownet_config.target`
resolves to the MI355X-native flow estimator (mgld_vsr_amd/raft.py)."""
from mgld_vsr_amd.raft import RAFT_SR  # noqa: F401
