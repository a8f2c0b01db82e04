#!/bin/bash
# r3: entries per XS tile against kernel time at several scales.  usage: tools/r3_tile_edges2.sh "<scale>:<te>,<te>,..." ...
for spec in "$@"; do
  scale=${spec%%:*}; tes=${spec#*:}
  B="python3 -u bench.py --scale $scale --steps 10 --warmup 2 --no-cpu-baseline --no-convergence --no-extra-configs"
  for te in ${tes//,/ }; do
    CUGRAPH_AMD_XS_TILE_EDGES=$te timeout -k 10 300 $B > gpurun_out/te.json 2> gpurun_out/te.err || { tail -5 gpurun_out/te.err; exit 1; }
    python3 -c "
import json
d=json.load(open('gpurun_out/te.json'))
print('scale $scale tile_edges %9d  ms/step %.4f xs2_kernel %.4f ms  plan %.3f s'%($te, d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['graph_build']['plan_build_s']))"
  done
done
