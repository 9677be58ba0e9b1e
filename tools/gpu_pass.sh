#!/bin/bash
# One GPU-box pass: parity tests, then A/B of the variant builds, then phase stamps.  Everything under gpurun_out/<tag>/.
# A step that is killed by its timeout ends the pass (no further GPU work after a hang).
tag=${1:-pass}; shift
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
run() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $out/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc"; tail -n ${TAILN:-6} $out/$name.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed: stopping"; exit 1; fi
  return $rc
}
run pytest 900 python -m pytest tests -q -m gpu --maxfail=${MAXFAIL:-6}
RANENV_SMALL_BATCH=0 run pytest_lean 900 python -m pytest tests -q -m gpu --maxfail=${MAXFAIL:-6}
for lm in ${LATE_MODES:-}; do RANENV_LATE=$lm run pytest_late$lm 900 python -m pytest tests -x -q -m gpu; done
if [ "$#" -gt 0 ]; then run ab 600 python tools/abprobe.py "$@"; fi
if [ -n "${STAMPS:-}" ] && [ -f tools/variants/stamps.so ]; then RANENV_LIB=$PWD/tools/variants/stamps.so run stamps 300 python tools/stamps.py; fi
exit 0
