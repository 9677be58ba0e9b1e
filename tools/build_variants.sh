#!/bin/bash
# Build A/B variants of the HIP library into tools/variants/<name>.so (run with RANENV_LIB=...).
# Usage: tools/build_variants.sh name1:"-DFLAG ..." name2:"..."
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -Wno-pass-failed -ffp-contract=off -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -disable-machine-licm \
    -I include $flags intent_radio_sched_multi_slice_amd/csrc/ranenv.hip -o tools/variants/$name.so
  echo "built tools/variants/$name.so ($flags)"
done
