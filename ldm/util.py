"""ldm/util.py:78-102 surface."""
from mgld_vsr_amd.util import default, exists, get_obj_from_str, instantiate_from_config  # noqa: F401
