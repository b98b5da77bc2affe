"""Drop-in import surface: the reference's `ldm.*` dotted paths (used as `target:` strings in its YAML configs,
configs/mgldvsr/mgldvsr_512_realbasicvsr_deg.yaml:4,35,55,80,88) resolve to the MI355X-native classes in mgld_vsr_amd."""
