#!/bin/bash
# scratch GPU session 8 (round 3): full GPU suite (no -x), smoke
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/g8_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > gpurun_out/g8_smoke.log
cat gpurun_out/g8_tests.log gpurun_out/g8_smoke.log; python - <<'PY'
import json
d=json.load(open('gpurun_out/parity_metrics.json'))
for k in sorted(d): print(f"{k:40s} {d[k]:.3e}")
PY
