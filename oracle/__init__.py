"""oracle/ — TEST INFRASTRUCTURE ONLY.

A CPU (torch fp32) restatement of the reference's algorithm for the MGLD-VSR hot path, function by function, each
citing the reference file:line it follows.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg may import it; the product path (mgld_vsr_amd/, ldm/, basicsr/, scripts/) never does.

Pinning: the reference ships no tests or golden vectors for this path.  The oracle is pinned against
outputs of the reference itself, imported in the build container by `tests/golden/make_golden.py`, committed as
small fixtures under `tests/golden/*.npz` and checked by `tests/test_oracle_golden.py`.
"""
