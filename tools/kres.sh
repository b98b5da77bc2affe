#!/bin/bash
# scratch: kernel resource table (VGPRs / spills / scratch / occupancy) of one HIP source   usage: tools/kres.sh igemm [filter]
cd "$(dirname "$0")/../mgld_vsr_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -Rpass-analysis=kernel-resource-usage \
  -c $1.hip -o /tmp/kres_$1.o 2>&1 | python3 /root/repo/tools/kres.py $2
