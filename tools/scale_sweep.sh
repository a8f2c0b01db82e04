#!/bin/bash
# iteration time at R-MAT scale 20 / 22 / 24 (the smaller BASELINE configs)
for sc in 20 22 24; do
  timeout -k 10 200 python3 bench.py --scale $sc --steps 20 --warmup 3 --no-cpu-baseline --no-convergence > gpurun_out/s$sc.json 2> gpurun_out/s$sc.err || { echo fail $sc; tail -3 gpurun_out/s$sc.err; exit 1; }
  python3 -c "import json;d=json.load(open('gpurun_out/s$sc.json'));print($sc, 'GTEPS', round(d['value'],1), 'ms', round(d['ms_per_step'],4), 'kern', round(d['roofline']['kernel_ms_avg'],4), d['roofline']['kernel'], 'whole-iter GB/s', round(d['achieved_hbm_GBps_whole_iteration']), d['buckets'])"
done
