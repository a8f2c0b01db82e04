#!/bin/bash
set -o pipefail
for cfg in "28 8 0" "28 8 0 private"; do
  CUGRAPH_AMD_MG_REGIONS=0 timeout -k 10 500 python3 -u tools/rank_emulate.py $cfg 2>&1 | grep -v "^$\|amdgpu.ids" | tee -a gpurun_out/r3_rank_emulate28.txt || exit 1
done
SCALES=28 bash tools/scale_sweep.sh | tee -a gpurun_out/r3_rank_emulate28.txt
