#!/bin/bash
# Register / LDS / spill figures of every kernel of the HIP library, as the compiler reports them (no GPU needed).
# Usage: tools/kstats.sh [extra hipcc flags]
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -Wno-pass-failed -ffp-contract=off \
  -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -disable-machine-licm -I include "$@" -Rpass-analysis=kernel-resource-usage \
  intent_radio_sched_multi_slice_amd/csrc/ranenv.hip -o /tmp/kstats.so 2>&1 | grep "remark:" | sed 's/ \[-Rpass.*//' |
  awk '/Function Name:/ {name=$NF} / VGPRs:/ {v=$NF} /TotalSGPRs:/ {s=$NF} /ScratchSize/ {sc=$NF} /VGPRs Spill/ {vs=$NF} /SGPRs Spill/ {ss=$NF} /Occupancy/ {o=$NF} /LDS Size/ {printf "%s VGPR %s SGPR %s scratch %s spillV %s spillS %s occ %s LDS %s\n", name, v, s, sc, vs, ss, o, $NF}' | c++filt | sed 's/(anonymous namespace):://g; s/(anonymous namespace)::KP//'
