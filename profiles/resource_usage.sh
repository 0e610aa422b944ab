#!/bin/bash
# kernel resource usage of one source file (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel
cd "$(dirname "$0")/../alevin-fry_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -x hip -c "$1" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "Function Name|VGPRs:|VGPRs Spill|SGPRs Spill|LDS Size|ScratchSize|Occupancy" | sed 's/.*remark: *//' |
  awk '/Function Name/ {if (l) print l; l=$3; next} {gsub(/ \[[^]]*\]/, ""); l=l "  " $0} END {print l}' | sed 's/  */ /g'
