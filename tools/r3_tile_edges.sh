#!/bin/bash
# r3: entries per XS tile (CUGRAPH_AMD_XS_TILE_EDGES) against xs2_kernel time.  usage: tools/r3_tile_edges.sh <scale> [ENV=VAL ...]
scale=$1; shift
B="python3 -u bench.py --scale $scale --steps 10 --warmup 2 --no-cpu-baseline --no-convergence --no-extra-configs"
for te in 16777216 33554432 67108864 134217728 268435456; do
  env "$@" CUGRAPH_AMD_XS_TILE_EDGES=$te timeout -k 10 200 $B > gpurun_out/te.json 2> gpurun_out/te.err || { tail -5 gpurun_out/te.err; exit 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/te.json'))
print('scale $scale tile_edges $te  ms/step %.4f xs2_kernel %.4f ms  plan %.3f s'%(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['graph_build']['plan_build_s']))"
done
