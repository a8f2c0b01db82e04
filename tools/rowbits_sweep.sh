#!/bin/bash
# rows per XS tile (2^bits) at the scales whose source space needs more than 16 windows under wide tiles
run() { sc=$1; shift
  env "$@" timeout -k 10 400 python3 bench.py --scale $sc --steps 5 --warmup 2 --no-cpu-baseline --no-convergence > gpurun_out/t.json 2> gpurun_out/t.err || { echo fail $sc "$@"; tail -3 gpurun_out/t.err; exit 1; }
  python3 -c "import json;d=json.load(open('gpurun_out/t.json'));print($sc, '$*', 'GTEPS', round(d['value'],1), 'ms', round(d['ms_per_step'],4), 'kern', round(d['roofline']['kernel_ms_avg'],4))"
}
for rb in 12 13 14; do run 28 CUGRAPH_AMD_XS_ROWBITS=$rb; done
for rb in 13 14; do run 27 CUGRAPH_AMD_XS_ROWBITS=$rb; done
run 26 A=1
