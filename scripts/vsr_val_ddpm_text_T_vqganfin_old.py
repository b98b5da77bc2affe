#!/usr/bin/env python
"""MI355X counterpart of scripts/vsr_val_ddpm_text_T_vqganfin_old.py (fixed input_size centre-crop entry, plain `model.sample`):
see mgld_vsr_amd/cli_simple.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from mgld_vsr_amd.cli_simple import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main(None, w_latent=False))
