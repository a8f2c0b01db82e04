#!/bin/bash
run() { sc=$1; te=$2
  CUGRAPH_AMD_XS_TILE_EDGES=$te timeout -k 10 200 python3 bench.py --scale $sc --steps 20 --warmup 3 --no-cpu-baseline --no-convergence > gpurun_out/t.json 2> gpurun_out/t.err || { echo fail $sc $te; tail -3 gpurun_out/t.err; exit 1; }
  python3 -c "import json;d=json.load(open('gpurun_out/t.json'));print($sc, $te, 'GTEPS', round(d['value'],1), 'ms', round(d['ms_per_step'],4), 'kern', round(d['roofline']['kernel_ms_avg'],4))"
}
for te in 32768 65536 131072 262144 524288 1048576; do run 22 $te; done
for te in 131072 262144 524288 1048576 2097152 4194304; do run 24 $te; done
for te in 16384 32768 65536 131072; do run 20 $te; done
