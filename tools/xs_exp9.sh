#!/bin/bash
run() { sc=$1; shift
  env "$@" timeout -k 10 200 python3 bench.py --scale $sc --steps 20 --warmup 3 --no-cpu-baseline --no-convergence > gpurun_out/t.json 2> gpurun_out/t.err || { echo fail $sc "$@"; tail -3 gpurun_out/t.err; exit 1; }
  python3 -c "import json;d=json.load(open('gpurun_out/t.json'));print($sc, '$*', 'GTEPS', round(d['value'],1), 'ms', round(d['ms_per_step'],4), 'kern', round(d['roofline']['kernel_ms_avg'],4))"
}
for sc in 20 22 24; do run $sc CUGRAPH_AMD_XS_ROWBITS=13; run $sc CUGRAPH_AMD_XS_ROWBITS=14; run $sc CUGRAPH_AMD_XS_ROWBITS=14 CUGRAPH_AMD_XS_PIECE=131072; done
