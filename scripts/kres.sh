#!/bin/bash
# kernel resource usage (VGPRs / spills / scratch) of one diagnostic TU: scripts/kres.sh build_diag/x.hip [extra flags]
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o "${src%.hip}.s" "$src" -Rpass-analysis=kernel-resource-usage "$@" 2>&1 \
  | grep -E "Function Name|VGPRs:|VGPRs Spill|ScratchSize" | sed -e 's/.*remark: [^ ]* //; s/\[-Rpass.*//' | paste - - - -
