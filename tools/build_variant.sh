#!/bin/bash
# scratch: build a variant of ONE translation unit into _variants/libmgld_<name>.so (kernel A/B runs: MGLD_HIP_LIB=_variants/libmgld_<name>.so)
#   usage: tools/build_variant.sh <name> <unit, e.g. conv3r> <extra hipcc flags...>
set -e
cd "$(dirname "$0")/.."
name=$1; unit=$2; shift; shift
mkdir -p _variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -Iinclude"
/opt/rocm/bin/hipcc $F "$@" -c mgld_vsr_amd/csrc/$unit.hip -o _variants/${unit}_$name.o
objs=""
for s in runtime igemm conv3q ppgemm conv3r pptconv norm attention elementwise raft hpenc; do
  if [ $s == $unit ]; then objs="$objs _variants/${unit}_$name.o"; else objs="$objs mgld_vsr_amd/csrc/$s.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _variants/libmgld_$name.so $objs
echo built _variants/libmgld_$name.so
