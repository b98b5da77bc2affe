"""mgld_vsr_amd — MI355X-native (gfx950) implementation of the MGLD-VSR per-segment inference hot path.

Layout: csrc/ (HIP kernels + C ABI, built into libmgld_hip.so), hip.py (ctypes binding), engine/ host mirror of
the reference's `ldm.*` plugin interface.  See DESIGN.md.
"""
__version__ = "0.1.0"
