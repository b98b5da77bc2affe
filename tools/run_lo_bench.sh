# A/B of the two-plane residual stream on the default bench workload (no CPU baseline / roofline legs)
for v in "unet,struct,vae_dec,vae_enc" "0" "vae_dec,vae_enc,struct" "unet,struct,vae_dec,vae_enc" "0"; do
  echo "MGLD_STREAM_LO=$v"
  MGLD_STREAM_LO=$v python bench.py --no-cpu-baseline --no-roofline --steps 6 --warmup 2 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('  value', d['value'], 'ms_per_step', d['ms_per_step'], 'one_at_a_time', d.get('value_one_at_a_time'))"
done
