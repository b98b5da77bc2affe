#!/bin/bash
# scratch: build an igemm variant into _variants/libmgld_<name>.so   usage: tools/build_variant.sh <name> <extra hipcc flags...>
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p _variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -Iinclude"
/opt/rocm/bin/hipcc $F "$@" -c mgld_vsr_amd/csrc/igemm.hip -o _variants/igemm_$name.o
objs=""
for s in runtime norm attention elementwise raft; do objs="$objs mgld_vsr_amd/csrc/$s.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _variants/libmgld_$name.so _variants/igemm_$name.o $objs
echo built _variants/libmgld_$name.so
