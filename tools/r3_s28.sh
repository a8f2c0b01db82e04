#!/bin/bash
# r3: scale 28 on one GPU: entries per lane and pass x entries per tile
B="python3 -u bench.py --scale 28 --steps 6 --warmup 2 --no-cpu-baseline --no-convergence --no-extra-configs"
for cfg in "8 16777216" "12 16777216" "16 16777216" "8 268435456" "12 268435456" "16 268435456"; do
  set -- $cfg
  CUGRAPH_AMD_XS_EPL=$1 CUGRAPH_AMD_XS_TILE_EDGES=$2 timeout -k 10 300 $B > gpurun_out/s28.json 2> gpurun_out/s28.err || { tail -5 gpurun_out/s28.err; exit 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/s28.json'))
print('scale 28 epl $1 tile_edges %9d  ms/step %.4f xs2_kernel %.4f ms  plan %.3f s'%($2, d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['graph_build']['plan_build_s']))"
done
