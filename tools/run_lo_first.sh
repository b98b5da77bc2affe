set -x
python -m pytest tests/test_stream_lo_gpu.py -x -q 2>&1 | tail -25
mkdir -p gpurun_out/lo_on gpurun_out/lo_off
python -m pytest tests/test_nets_gpu.py -q -k "structcond_small or unet_small or vae_small or vae_decoder or unet_fullwidth or config0_fullwidth or pipeline_fullwidth" 2>&1 | tail -15
cp gpurun_out/parity_metrics.json gpurun_out/lo_on/
MGLD_STREAM_LO=0 python -m pytest tests/test_nets_gpu.py -q -k "structcond_small or unet_small or vae_small or vae_decoder or unet_fullwidth or config0_fullwidth or pipeline_fullwidth" 2>&1 | tail -5
cp gpurun_out/parity_metrics.json gpurun_out/lo_off/
python - <<'PY'
import json
a=json.load(open('gpurun_out/lo_on/parity_metrics.json')); b=json.load(open('gpurun_out/lo_off/parity_metrics.json'))
for k in sorted(a):
    if k in b: print(f"{k:40s} lo_on {a[k]:.3e}  lo_off {b[k]:.3e}  ratio {a[k]/max(b[k],1e-30):.3f}")
PY
