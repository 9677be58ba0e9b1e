#!/bin/bash
# Build A/B / diagnostic variants of the HIP library into tools/variants/<name>.so (run with RANENV_LIB=...).
# Usage: tools/build_variants.sh name1:"-DFLAG ..." name2:"..."      (the -D flags go to every object of the library)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  python3 -m intent_radio_sched_multi_slice_amd.csrc.build -o tools/variants/$name.so $flags > /dev/null
  echo "built tools/variants/$name.so ($flags)"
done
