#!/bin/bash
# r3: the personalised exchange on HIP sessions (ranks sharing GPU 0 over gloo), then one rank of 8 with shared and with private slots
set -o pipefail
timeout -k 10 900 python3 -u -m pytest tests/test_gpu_mg_world.py -m gpu -x -q --timeout 200 --timeout-method thread -k "sharing" > gpurun_out/r3_mg_world.txt 2>&1 || { tail -40 gpurun_out/r3_mg_world.txt; exit 1; }
tail -2 gpurun_out/r3_mg_world.txt
for cfg in "26 8 0" "26 8 0 private" "26 4 0" "26 4 0 private" "26 2 0" "26 2 0 private"; do
  CUGRAPH_AMD_MG_REGIONS=0 timeout -k 10 300 python3 -u tools/rank_emulate.py $cfg 2>&1 | grep -v "^$" | tee -a gpurun_out/r3_rank_emulate.txt || exit 1
done
