#!/bin/bash
# Counter pass over several builds (tools/variants/<name>.so): tools/pmc_variants.sh <tag> "<counters>" name ...
tag=$1; ctrs=$2; shift 2
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for v in "$@"; do
  export RANENV_LIB=$PWD/tools/variants/$v.so
  timeout -k 10 300 rocprofv3 --pmc $ctrs -d $out/$v -o p --output-format csv -- python3 tools/profile_step.py 30 > $out/$v.log 2>&1; rc=$?
  echo "[$v] rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
  python3 tools/pmc_summary.py $out/$v | grep core_kernel | sed "s/^/$v /" | tee -a $out/summary.txt
done
