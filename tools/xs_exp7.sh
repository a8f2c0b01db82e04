#!/bin/bash
run() { sc=$1; te=$2; pc=$3
  CUGRAPH_AMD_XS_TILE_EDGES=$te CUGRAPH_AMD_XS_PIECE=$pc timeout -k 10 200 python3 bench.py --scale $sc --steps 20 --warmup 3 --no-cpu-baseline --no-convergence > gpurun_out/t.json 2> gpurun_out/t.err || { echo fail $sc $te; tail -3 gpurun_out/t.err; exit 1; }
  python3 -c "import json;d=json.load(open('gpurun_out/t.json'));print($sc, $te, $pc, 'GTEPS', round(d['value'],1), 'ms', round(d['ms_per_step'],4), 'kern', round(d['roofline']['kernel_ms_avg'],4))"
}
for pc in 32768 65536 131072 262144; do for te in 1048576 4194304 16777216; do run 22 $te $pc; done; done
for pc in 65536 131072 262144; do for te in 2097152 8388608 33554432; do run 24 $te $pc; done; done
for pc in 16384 32768 65536; do for te in 262144 1048576 4194304; do run 20 $te $pc; done; done
