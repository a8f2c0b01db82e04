#!/bin/bash
# iteration time on small graphs (launch-bound regime): R-MAT scale 10 .. 20
for sc in ${@:-10 12 14 16 18 20}; do
  timeout -k 10 200 python3 bench.py --scale $sc --steps 200 --warmup 20 --no-cpu-baseline --no-convergence > gpurun_out/s$sc.json 2> gpurun_out/s$sc.err || { echo fail $sc; tail -3 gpurun_out/s$sc.err; exit 1; }
  python3 -c "import json;d=json.load(open('gpurun_out/s$sc.json'));print($sc, 'GTEPS', round(d['value'],2), 'us/iter', round(d['ms_per_step']*1e3,2), 'dominant kernel us', round(d['roofline']['kernel_ms_avg']*1e3,2), d['roofline']['kernel'], 'event us', round(d['roofline']['iteration_ms_event_avg']*1e3,2))"
done
