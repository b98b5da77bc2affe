#!/bin/bash
# One parametrised runner for the GPU leases of a round (replaces the one-off g<N>.sh scripts of rounds 1-3):
#   gpurun --timeout T -- 'bash tools/lease.sh <recipe> [args...]'
# Every recipe writes under gpurun_out/ (merged back by gpurun) and prints a short summary last.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
recipe=$1; shift
case "$recipe" in
  pp_lin)     # bring-up of the ping-pong LINEAR kernels: lane-swap probe, unit tests, per-shape A/B against the 128-class kernels
    ./_variants/pl16_probe 2>&1 | tail -18 | tee gpurun_out/pl16_probe.log
    timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "pingpong" 2>&1 | tail -15 | tee gpurun_out/pp_tests.log
    timeout 900 python tools/igemm_bench.py lin --nst ${1:-12,20,21,22,23,24,25,26,27} --rounds 2 --json gpurun_out/pp_lin.json 2>&1 | tee gpurun_out/pp_lin.log | cut -c1-400
    ;;
  pp_conv)    # bring-up of the ping-pong patch convolutions: unit tests, per-shape A/B against conv3q
    timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "pingpong" 2>&1 | tail -15 | tee gpurun_out/pp_tests.log
    timeout 900 python tools/igemm_bench.py conv --variants ${1:-0,31,32,33,34,35,36,37,38,39} --rounds 2 --json gpurun_out/pp_conv.json 2>&1 | tee gpurun_out/pp_conv.log | cut -c1-420
    timeout 900 python tools/igemm_bench.py vae --variants ${1:-0,31,32,33,34,35,36,37,38,39} --rounds 2 --json gpurun_out/pp_vae.json 2>&1 | tee gpurun_out/pp_vae.log | cut -c1-420
    ;;
  ablate)     # timing of ablation builds (_variants/libmgld_<name>.so, tools/build_variant.sh): <bench group> <only> <tunes> <names...>
    what=$1; only=$2; tunes=$3; shift; shift; shift
    for n in base "$@"; do
      lib=""; [ $n != base ] && lib=_variants/libmgld_$n.so
      echo "== $n"; MGLD_HIP_LIB=$lib timeout 300 python tools/igemm_bench.py $what --only "$only" --variants $tunes --nst $tunes --rounds 2 2>&1 | grep -v amdgpu.ids | cut -c1-300
    done | tee gpurun_out/ablate.log
    ;;
  c3split)    # conv3r K split on the 16^2 level: unit tests + A/B against conv3q's split (tune 0 = planner, 5 = conv3q variant, 54/55/58 = conv3r cfg 4/5/8 split)
    timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "ksplit or frame_stacked or conv3x3_pingpong" 2>&1 | tail -6 | tee gpurun_out/c3split_tests.log
    timeout 600 python tools/igemm_bench.py conv --only conv16 --variants ${1:-0,5,54,55,58,35,39} --rounds 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c3split.log | cut -c1-900
    MGLD_CONV3R_W8=0 timeout 600 python tools/igemm_bench.py conv --only conv8 --variants 0 --rounds 3 2>&1 | grep -v amdgpu.ids | grep conv8 | sed "s/^/conv3q planner: /" | tee gpurun_out/c3w8.log | cut -c1-400
    timeout 600 python tools/igemm_bench.py conv --only conv8 --variants 0,59,60,40,41 --rounds 3 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/c3w8.log | cut -c1-900
    ;;
  tconv)      # the temporal Conv3d: unit tests + microbench (implicit-GEMM kernel in both row orders against the ping-pong kernel)
    timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "tconv" 2>&1 | tail -15 | tee gpurun_out/tconv_tests.log
    timeout 600 python tools/tconv_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/tconv.log | cut -c1-400
    ;;
  attn)       # attention microbench under the given env settings (one per argument), e.g. "MGLD_ATTN_DMA=0"
    for envs in "" "$@"; do
      echo "== ${envs:-default}"; env $envs timeout 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids
    done | tee gpurun_out/attn.log
    ;;
  tests)      # the GPU suite (optionally -k <expr>)
    timeout 2400 python -m pytest tests -m gpu -q -x "$@" 2>&1 | tail -15 | tee gpurun_out/tests.log
    ;;
  ab)         # end-to-end A/B: bench.py --inflight 1 (one segment at a time) under the env settings given as arguments, alternated twice
    for rep in 1 2 3; do for envs in "$@"; do
      tag=$(echo "$envs" | tr ' =' '__')
      env $envs timeout 300 python bench.py --inflight 1 --steps ${AB_STEPS:-6} --warmup 2 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab_${tag}_$rep.json
      python -c "import json;d=json.load(open('gpurun_out/ab_${tag}_$rep.json'));print('$envs rep $rep:',d['value'],d['ms_per_step'])"
    done; done | tee gpurun_out/ab.log
    ;;
  bench)      # bench.py with the given flags; the JSON line goes to gpurun_out/bench_<tag>.json
    tag=$1; shift
    timeout 1200 python bench.py "$@" 2> gpurun_out/bench_$tag.err | tail -1 > gpurun_out/bench_$tag.json
    cut -c1-600 gpurun_out/bench_$tag.json
    ;;
  prof)       # rocprofv3 --kernel-trace --stats of the default bench workload -> gpurun_out/kernel_stats.txt (+ .json)
    bash tools/prof_run.sh
    ;;
  pmc)        # HBM traffic per launch (FETCH_SIZE / WRITE_SIZE in separate passes) -> gpurun_out/pmc_traffic.json
    bash tools/pmc_traffic.sh
    ;;
  r5a)        # round 5, first contact: fp32 RAFT kernels + network, clip batching, ragged GroupNorm windows; then clips vs in-flight A/B; then the x0 probe
    timeout 900 python -m pytest tests/test_sampler_kernels_gpu.py tests/test_kernels_gpu.py tests/test_nets_gpu.py -q -x \
      -k "conv_f32 or instnorm or corr_lookup or gru_and or groupnorm_stats_of_output or raft or estimate_flows or clips" 2>&1 | tail -25 | tee gpurun_out/r5a_tests.log
    timeout 600 python -m pytest tests/test_cli_gpu.py -q -x -k "fixed_size_cli_reproduces_the_reference_script_at_the_production_schedule" 2>&1 | tail -25 | tee gpurun_out/r5a_cli.log
    for cfg in "--inflight 3" "--clips 3 --inflight 1" "--clips 2 --inflight 2" "--clips 2 --inflight 1" "--inflight 1"; do
      tag=$(echo "$cfg" | tr -d ' -')
      timeout 400 python bench.py $cfg --steps 6 --warmup 2 --no-roofline --no-cpu-baseline --no-one-at-a-time 2> gpurun_out/r5a_bench_$tag.err | tail -1 > gpurun_out/r5a_bench_$tag.json
      python -c "import json;d=json.load(open('gpurun_out/r5a_bench_$tag.json'));print('$cfg:',d['value'],'fps',d['ms_per_step'],'ms/step')" 2>&1 | tee -a gpurun_out/r5a_bench.log
    done
    for c in "$@"; do timeout 600 python tools/x0_probe.py $c 50 "default" 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a gpurun_out/r5a_probe.log; done
    ;;
  r5b)        # round 5: smooth-workload error decomposition, more clips x in-flight combinations, per-kernel table at clips = 2, text tower
    timeout 900 python tools/x0_probe.py c2s 50 "default;all" 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/r5b_probe.log
    timeout 600 python -m pytest tests/test_nets_gpu.py tests/test_workloads_gpu.py -q -k "text_tower or c2s or clips" 2>&1 | tail -15 | tee gpurun_out/r5b_tests.log
    for cfg in "--clips 3 --inflight 2" "--clips 2 --inflight 3" "--clips 4 --inflight 2"; do
      tag=$(echo "$cfg" | tr -d ' -')
      timeout 500 python bench.py $cfg --steps 6 --warmup 2 --no-roofline --no-cpu-baseline --no-one-at-a-time 2> gpurun_out/r5b_bench_$tag.err | tail -1 > gpurun_out/r5b_bench_$tag.json
      python -c "import json;d=json.load(open('gpurun_out/r5b_bench_$tag.json'));print('$cfg:',d['value'],'fps',d['ms_per_step'],'ms/step')" 2>&1 | tee -a gpurun_out/r5b_bench.log
    done
    timeout 600 python bench.py --clips 2 --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --dump-shapes gpurun_out/r5b_shapes_clips2.json 2> gpurun_out/r5b_roof.err | tail -1 > gpurun_out/r5b_roof_clips2.json
    python -c "
import json;d=json.load(open('gpurun_out/r5b_roof_clips2.json'))
print(d['value'],d['ms_per_step'])
for e in d['roofline']['by_kernel']: print(e['kernel'][:70],e['ms_per_segment'],e['launches'],e['frac'])
print({k:(v['ms_per_segment'],v['launches'],v['frac_of_peak']) for k,v in d['roofline']['hbm']['kernels'].items()})" 2>&1 | tee gpurun_out/r5b_roof.log
    ;;
  r5c)        # round 5: the high-precision first-stage encoder: kernel tests, encoder vs oracle, smooth workload, the production-schedule harness runs
    timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_nets_gpu.py -q -x -k "hp_ or vae_small or conv_f32" 2>&1 | tail -15 | tee gpurun_out/r5c_tests.log
    timeout 900 python tools/x0_probe.py c2s 50 "default" 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/r5c_probe.log
    timeout 1200 python -m pytest tests/test_workloads_gpu.py -q -k "c2s or c2-50 or c2-4" 2>&1 | tail -15 | tee gpurun_out/r5c_work.log
    timeout 900 python -m pytest tests/test_cli_gpu.py -q -k "production_schedule" 2>&1 | tail -15 | tee gpurun_out/r5c_cli.log
    for cfg in "--inflight 1" "--clips 2 --inflight 2"; do
      tag=$(echo "$cfg" | tr -d ' -')
      timeout 400 python bench.py $cfg --steps 6 --warmup 2 --no-roofline --no-cpu-baseline --no-one-at-a-time 2> gpurun_out/r5c_bench_$tag.err | tail -1 > gpurun_out/r5c_bench_$tag.json
      python -c "import json;d=json.load(open('gpurun_out/r5c_bench_$tag.json'));print('$cfg:',d['value'],'fps',d['ms_per_step'],'ms/step')" 2>&1 | tee -a gpurun_out/r5c_bench.log
    done
    ;;
  hpprof)     # the high-precision first-stage encode: time vs the fp16 encoder, per-kernel split under rocprofv3
    for hp in 1 0; do MGLD_HP_ENCODER=$hp timeout 300 python tools/hp_bench.py 8 16 2>&1 | grep HP_ENCODER | tee -a gpurun_out/hp_bench.log; done
    R=$GRAFT_REPO_ROOT; rm -rf $R/gpurun_out/hpprof; mkdir -p $R/gpurun_out/hpprof
    ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/hpprof -o hp -- python $R/tools/hp_bench.py 8 > /dev/null 2>&1 )
    python tools/kstats.py gpurun_out/hpprof 2>/dev/null | head -30 | cut -c1-200 | tee gpurun_out/hp_kstats.txt
    find gpurun_out/hpprof -name "*.csv" -size +1M -delete
    ;;
  r5e)        # round 5: 64-query-row attention waves + the ping-pong kernel's fp32 epilogue: tests, microbenches, end to end
    timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "flash or attention or hp_" 2>&1 | tail -8 | tee gpurun_out/r5e_tests.log
    for qh in 1 2; do echo "== MGLD_ATTN_QH=$qh"; MGLD_ATTN_QH=$qh timeout 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r5e_attn.log
    timeout 300 python tools/hp_bench.py 8 2>&1 | grep HP_ENCODER | tee gpurun_out/r5e_hp.log
    for envs in "MGLD_ATTN_QH=1" "MGLD_ATTN_QH=2"; do
      env $envs timeout 400 python bench.py --steps 6 --warmup 2 --no-roofline --no-cpu-baseline --no-one-at-a-time 2>/dev/null | tail -1 > gpurun_out/r5e_bench_$envs.json
      python -c "import json;d=json.load(open('gpurun_out/r5e_bench_$envs.json'));print('$envs:',d['value'],'fps',d['ms_per_step'],'ms/step')" 2>&1 | tee -a gpurun_out/r5e_bench.log
    done
    ;;
  r5g)        # round 5: pre-scaled-query attention (max carried by the contraction): tests, microbench A/B, end-to-end A/B, parity of the workloads
    timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "flash or attention or tile_conv3p" 2>&1 | tail -8 | tee gpurun_out/r5g_tests.log
    for ps in 0 1; do echo "== ATTN_BENCH_PS=$ps"; ATTN_BENCH_PS=$ps timeout 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r5g_attn.log
    timeout 1200 python -m pytest tests/test_workloads_gpu.py tests/test_nets_gpu.py -q -k "c2-50 or c2s-50 or c2-4 or unet_full or c1 or e2e_full" 2>&1 | tail -8 | tee gpurun_out/r5g_work.log
    for envs in "MGLD_ATTN_PS=0" "MGLD_ATTN_PS=1"; do
      env $envs timeout 400 python bench.py --steps 6 --warmup 2 --no-roofline --no-cpu-baseline --no-one-at-a-time 2>/dev/null | tail -1 > gpurun_out/r5g_bench_$envs.json
      python -c "import json;d=json.load(open('gpurun_out/r5g_bench_$envs.json'));print('$envs:',d['value'],'fps',d['ms_per_step'],'ms/step')" 2>&1 | tee -a gpurun_out/r5g_bench.log
    done
    ;;
  *) echo "unknown recipe $recipe"; exit 2 ;;
esac
