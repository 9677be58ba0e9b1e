#!/bin/bash
# One measurement pass on the GPU box: parity tests, smoke, PMC traffic, rocprofv3 kernel stats, bench line.
# Usage (through gpurun): bash tools/round_measure.sh <tag>     -> everything lands in gpurun_out/<tag>/
set -o pipefail
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1 || { tail -20 $out/pytest_gpu.log; exit 1; }
tail -2 $out/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1 || { tail -20 $out/smoke.log; exit 1; }
tail -1 $out/smoke.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c -d $out/pmc_$c -o p --output-format csv -- python tools/profile_step.py 30 > $out/pmc_$c.log 2>&1 || { tail -5 $out/pmc_$c.log; exit 1; }
done
python tools/pmc_traffic.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE 4096 2 $out/pmc_traffic.json && cp $out/pmc_traffic.json profiles/pmc_traffic.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/prof -o p --output-format csv -- python bench.py --steps 200 --no-cpu-baseline > $out/bench_prof.log 2>&1 || { tail -5 $out/bench_prof.log; exit 1; }
timeout -k 10 600 python bench.py > $out/bench.log 2>&1 || { tail -20 $out/bench.log; exit 1; }
tail -1 $out/bench.log
