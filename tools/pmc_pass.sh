#!/bin/bash
# rocprofv3 counter passes over tools/profile_step.py (counters only: no trace options).  Usage: tools/pmc_pass.sh <tag> "<ctr ctr ...>" ["<ctr ...>" ...]
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set -d $out/pmc$i -o p --output-format csv -- python3 tools/profile_step.py 30 > $out/pmc$i.log 2>&1; rc=$?
  echo "[pmc$i: $set] rc=$rc"; tail -2 $out/pmc$i.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
python3 tools/pmc_summary.py $out/pmc* | tee $out/pmc_summary.txt
