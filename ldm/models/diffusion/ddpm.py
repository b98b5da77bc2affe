"""`ldm.models.diffusion.ddpm` surface (reference ddpm.py:101-154, 3166-4940) -> mgld_vsr_amd.ddpm."""
from mgld_vsr_amd.ddpm import (DiffusionWrapper, LatentDiffusionVSRTextWT, extract_into_tensor, make_beta_schedule,  # noqa: F401
                               space_timesteps)
