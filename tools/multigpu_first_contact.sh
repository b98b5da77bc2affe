#!/bin/bash
# First contact with a multi-GPU node (none was available to rounds 1-4: SCALE_r0x.json "skipped").  What the driver would run,
# unchanged:   bash tools/multigpu_first_contact.sh [max_gpus]
#   0. a 60-second gate: two ranks over RCCL through the same DistComm calls the sharded modes make (launcher, rendezvous, transport);
#   1. for every mode — segment-parallel (weak scaling, no data-path collective), frame-sharded, tile-sharded (strong scaling of ONE
#      segment: halo send/recv + all-gathers over xGMI) — bench.py --gpus N for N = 1, 2, 4, 8 (up to the GPUs present);
#   2. ONE SCALE-shaped JSON line per mode: per N the bench line's value / ms_per_step, plus, for the sharded modes, comm_plan()'s
#      predicted bytes next to the measured bytes and exchange time of rank 0.
# Results also land in gpurun_out/scale_<mode>.json.
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONUNBUFFERED=1
mkdir -p gpurun_out
NG=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
MAX=${1:-$NG}; [ "$MAX" -gt "$NG" ] && MAX=$NG
echo "[first contact] $NG GPUs visible, using up to $MAX"
if [ "$MAX" -ge 2 ]; then
  timeout 60 python bench.py --gpus 2 --backend nccl --spawn-selftest --frame-shard > gpurun_out/nccl_gate.json 2> gpurun_out/nccl_gate.err \
    || { echo "[first contact] RCCL gate FAILED (gpurun_out/nccl_gate.err):"; tail -5 gpurun_out/nccl_gate.err; exit 3; }
  echo "[first contact] RCCL gate ok: $(cut -c1-300 gpurun_out/nccl_gate.json)"
fi
STEPS=${STEPS:-4}; WARM=${WARM:-1}
for mode in segment frame tile; do
  case $mode in
    segment) flags="" ;;
    frame)   flags="--frame-shard" ;;
    tile)    flags="--tile --tile-shard --size 1024 --frames 4 --guidance" ;;
  esac
  : > gpurun_out/scale_$mode.lines
  for n in 1 2 4 8; do
    [ $n -gt $MAX ] && continue
    timeout 1500 python bench.py --gpus $n --backend nccl --steps $STEPS --warmup $WARM --no-roofline --no-cpu-baseline --no-one-at-a-time $flags \
      2> gpurun_out/scale_${mode}_$n.err | tail -1 >> gpurun_out/scale_$mode.lines || echo "{\"n_gpus\": $n, \"error\": \"bench.py failed, see gpurun_out/scale_${mode}_$n.err\"}" >> gpurun_out/scale_$mode.lines
  done
  python - "$mode" <<'PY'
import json, sys
mode = sys.argv[1]
runs = []
for ln in open(f"gpurun_out/scale_{mode}.lines"):
    ln = ln.strip()
    if not ln:
        continue
    try:
        d = json.loads(ln)
    except ValueError:
        continue
    if "error" in d:
        runs.append(d)
        continue
    runs.append({"n_gpus": d["n_gpus"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "scaling": d["scaling"],
                 "parallelism": d["config"]["parallelism"], "segments_in_flight": d["config"]["segments_in_flight"],
                 "graphs_per_step": d["config"]["graphs_per_step"], "comm": d["config"].get("comm")})
base = next((r["value"] for r in runs if r.get("n_gpus") == 1 and "value" in r), None)
line = {"metric": "HR frames/sec at 512^2, 50 DDPM steps" if mode != "tile" else "HR frames/sec at 1024^2 (aggregation sampling), 50 DDPM steps",
        "mode": mode, "runs": runs, "speedup_vs_1": {str(r["n_gpus"]): round(r["value"] / base, 3) for r in runs if base and "value" in r}}
json.dump(line, open(f"gpurun_out/scale_{mode}.json", "w"))
print(json.dumps(line))
PY
done
