#!/bin/bash
# Register / LDS / occupancy figures of the library's kernels as the compiler reports them (device-only compile, no GPU needed):
#   tools/kernel_regs.sh [name-pattern]
cd "$(dirname "$0")/.." || exit 1
pat="${1:-.}"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
  -Iinclude -Iadapted_amd/csrc --cuda-device-only -c -o /dev/null -Rpass-analysis=kernel-resource-usage adapted_amd/csrc/adapted_hip.hip 2>&1 |
  grep -E "Function Name|VGPRs:|AGPRs|SGPRs:|Occupancy|LDS Size|ScratchSize" |
  awk -v pat="$pat" '/Function Name/ {show = ($0 ~ pat)} show {sub(/^.*remark: [^ ]* /, ""); print}'
