#!/bin/bash
# One measurement pass on the GPU box (round 6): parity tests, bench lines, PMC passes of the schedules bench.py times (headline, gather,
# native size, configs[1]), rocprofv3 kernel stats, the K = 20 block and the trainer-facing schedules' timelines.
# Usage (through gpurun): bash tools/round6_measure.sh <tag> [skip_pytest] [part]   -> gpurun_out/<tag>/      part: all (default: bench + pmc + prof) | bench | pmc | prof | phases
set -o pipefail
tag=${1:-r06}; part=${3:-all}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
step() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $out/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc"; grep -v amdgpu.ids $out/$name.log | tail -n ${TAILN:-2} | cut -c1-400
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit 1; fi
}
if [ -z "$2" ] || [ "$2" = tests ]; then step pytest_gpu 900 python3 -m pytest tests -q -m gpu; fi
if [ $part = all ] || [ $part = bench ]; then
step bench 900 python3 bench.py
step bench_driver 300 python3 bench.py --gpus 1 --steps 20 --warmup 5
for K in 10 100; do step bench_k$K 300 python3 bench.py --steps $K --warmup $((K/4 + 1)) --no-cpu-baseline --no-other-configs; done
RANENV_PERSIST=1 step bench_persist 600 python3 bench.py --no-cpu-baseline --no-other-configs
step bench_cfg1 300 python3 bench.py --config 1 --no-cpu-baseline
step bench_cfg4 300 python3 bench.py --config 4 --no-cpu-baseline
step bench_native 400 python3 bench.py --config native --no-cpu-baseline
step bench_philox 300 python3 bench.py --traffic philox --no-cpu-baseline --no-gather --no-other-configs
step rehearse2 400 python3 bench.py --gpus 2 --rehearse-on-one-gpu --traces 40 --trace-len 100 --no-cpu-baseline --steps 20 --warmup 5
fi
if [ $part = all ] || [ $part = pmc ]; then
K=100; CALLS=2; TT=$((10 + K * CALLS))
pmc() {  # name mode config
  local n=$1 mode=$2 cfg=$3
  for c in FETCH_SIZE WRITE_SIZE; do
    step pmc_${n}_$c 300 rocprofv3 --pmc $c -d $out/pmc_${n}_$c -o p --output-format csv -- python3 tools/profile_rollout.py $K $CALLS $mode $cfg
  done
  step pmc_${n}_sq 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $out/pmc_${n}_sq -o p --output-format csv -- python3 tools/profile_rollout.py $K $CALLS $mode $cfg
  step pmc_${n}_sq2 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $out/pmc_${n}_sq2 -o p --output-format csv -- python3 tools/profile_rollout.py $K $CALLS $mode $cfg
}
pmc stream stream 2; pmc gather gather 2; pmc native stream 5; pmc config1 stream 1
d() { echo $out/pmc_$1_FETCH_SIZE,$out/pmc_$1_WRITE_SIZE,$out/pmc_$1_sq,$out/pmc_$1_sq2; }
python3 tools/pmc_collect_r4.py $out/r06_pmc.json 4096 2 stream_rollout:$TT:$(d stream) gather_rollout:$TT:$(d gather) \
  native_rollout:$TT:$(d native):16384:5 config1_rollout:$TT:$(d config1):1024:1 > $out/pmc_collect.log
python3 tools/pmc_summary.py $out/pmc_stream_sq $out/pmc_stream_sq2 > $out/pmc_stream_summary.txt
python3 tools/pmc_summary.py $out/pmc_gather_sq $out/pmc_gather_sq2 > $out/pmc_gather_summary.txt
python3 tools/pmc_summary.py $out/pmc_native_sq $out/pmc_native_sq2 > $out/pmc_native_summary.txt
python3 tools/pmc_summary.py $out/pmc_config1_sq $out/pmc_config1_sq2 > $out/pmc_config1_summary.txt
fi
if [ $part = all ] || [ $part = prof ]; then
step prof_stream 400 rocprofv3 --kernel-trace --stats -d $out/prof_stream -o p --output-format csv -- python3 bench.py --steps 200 --no-cpu-baseline --no-single-stream --no-gather --no-other-configs
step prof_stream_k20 400 rocprofv3 --kernel-trace --stats -d $out/prof_stream_k20 -o p --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-single-stream --no-gather --no-other-configs
step prof_gather 400 rocprofv3 --kernel-trace --stats -d $out/prof_gather -o p --output-format csv -- python3 bench.py --steps 200 --no-cpu-baseline --only-gather --no-other-configs
step prof_steploop 400 rocprofv3 --kernel-trace -d $out/prof_steploop -o p --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gather --no-other-configs
python3 tools/steploop_timeline.py $out/prof_steploop/p_kernel_trace.csv 20 > $out/steploop_timeline.txt 2>&1 || python3 tools/steploop_timeline.py $(ls $out/prof_steploop/*/p_kernel_trace.csv | head -1) 20 > $out/steploop_timeline.txt 2>&1
cat $out/steploop_timeline.txt
fi
if [ $part = phases ]; then      # (needs tools/variants/diag{3,4,7,11,12}.so: tools/build_variants.sh in the build container)
RANENV_LIB=$PWD/tools/variants/diag12.so RANENV_PERSIST=1 step phases_stream 300 python3 tools/persist_phases.py 2 200
RANENV_LIB=$PWD/tools/variants/diag12.so step phases_gather 300 python3 tools/persist_phases.py 2 200 gather
RANENV_LIB=$PWD/tools/variants/diag12.so step phases_config1 300 python3 tools/persist_phases.py 1 200
TAILN=12 step valu_phases 1200 bash tools/valu_phases.sh $out/valu_phases
fi
echo "pass complete"
