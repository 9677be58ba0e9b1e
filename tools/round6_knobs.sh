#!/bin/bash
# The GPU suite under the experiment knobs (every option changes a schedule or a build, never a result).
tag=${1:-r06knobs}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
run() {  # name env...
  local name=$1; shift
  env "$@" timeout -k 10 600 python3 -m pytest tests -q -m gpu -x > $out/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc $(grep -E 'passed|failed' $out/$name.log | tail -1)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit 1; fi
}
run compact0 RANENV_COMPACT=0
run row16 RANENV_ROW_WIDTH=16
run fuse3 RANENV_FUSE=3
run fuse1 RANENV_FUSE=1
run fuse20 RANENV_FUSE=20
run small0 RANENV_SMALL_BATCH=0
run persist1_chunk2 RANENV_PERSIST=1 RANENV_PERSIST_CHUNK=2
run persist0_mix0_pack0 RANENV_PERSIST=0 RANENV_MIX=0 RANENV_PACK=0
run gather_row16_fuse20_compact0 RANENV_SE_MODE=gather RANENV_ROW_WIDTH=16 RANENV_FUSE=20 RANENV_COMPACT=0
run persist_handover_stress RANENV_PERSIST=1 RANENV_PERSIST_GRID=64 RANENV_PERSIST_CHUNK=1
run rbmajor_persist1 RANENV_SE_LAYOUT=rb RANENV_PERSIST=1
run rbmajor_gather RANENV_SE_LAYOUT=rb RANENV_SE_MODE=gather
run shortcut0 RANENV_AUTORESET_SHORTCUT=0
echo "pass complete"
