#!/bin/bash
# scratch: run tools/attn_sp_check.py once per kernel variant (the library reads MGLD_ATTN_* once per process); output -> gpurun_out/attn_sp_sweep.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/attn_sp_sweep.txt
: > $out
for v in ${@:-0 1 2 3 4}; do
  echo "=== MGLD_ATTN_SP=$v" >> $out
  MGLD_ATTN_SP=$v timeout 300 python tools/attn_sp_check.py >> $out 2>&1
  echo "rc=$?" >> $out
done
cat $out
