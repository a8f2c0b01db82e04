#!/bin/bash
# iteration and build time at R-MAT scale 20 / 22 / 24 / 28 (the other BASELINE sizes; 28: 64-bit offsets, one GPU)
for sc in ${SCALES:-20 22 24 28}; do
  timeout -k 10 400 python3 bench.py --scale $sc --steps 20 --warmup 3 --no-cpu-baseline --no-convergence --no-extra-configs > gpurun_out/s$sc.json 2> gpurun_out/s$sc.err || { echo fail $sc; tail -3 gpurun_out/s$sc.err; exit 1; }
  python3 -c "import json;d=json.loads(open('gpurun_out/s$sc.json').read().strip().splitlines()[-1]);print($sc, 'GTEPS', round(d['value'],1), 'ms', round(d['ms_per_step'],4), 'kern', round(d['roofline']['kernel_ms_avg'],4), d['roofline']['kernel'], 'frac', round(d['roofline']['frac'],3), 'build', round(d['graph_build']['coo_to_csc_s'],3), 'plan', round(d['graph_build']['plan_build_s'],3))"
done
