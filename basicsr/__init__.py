"""Drop-in import surface for the two basicsr pieces the VSR inference path touches (arch_util, raft_arch)."""
