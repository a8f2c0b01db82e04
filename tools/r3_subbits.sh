#!/bin/bash
# r3: plan build time against sort sub-units for big tiles.  usage: tools/r3_subbits.sh <scale> <tile_edges> <subbits> ...
scale=$1; te=$2; shift 2
B="python3 -u bench.py --scale $scale --steps 6 --warmup 2 --no-cpu-baseline --no-convergence --no-extra-configs"
for sb in "$@"; do
  CUGRAPH_AMD_XS_SORT_SUBBITS=$sb CUGRAPH_AMD_XS_TILE_EDGES=$te timeout -k 10 300 $B > gpurun_out/sb.json 2> gpurun_out/sb.err || { tail -5 gpurun_out/sb.err; exit 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/sb.json'))
print('scale $scale tile_edges %9d subbits $sb  ms/step %.4f dominant kernel %.4f ms  plan %.3f s'%($te, d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['graph_build']['plan_build_s']))"
done
