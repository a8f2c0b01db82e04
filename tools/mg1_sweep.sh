#!/bin/bash
# the multi-GPU driver on ONE rank (collectives on): with and without the two-region layout
run() { name=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py --force-mg --steps 20 --warmup 3 --no-cpu-baseline --no-convergence > gpurun_out/mg_$name.json 2> gpurun_out/mg_$name.err || { echo fail $name; tail -5 gpurun_out/mg_$name.err; exit 1; }
  python3 -c "import json;d=json.load(open('gpurun_out/mg_$name.json'));print('$name', 'GTEPS', round(d['value'],1), 'ms', round(d['ms_per_step'],4), 'kern', round(d['roofline']['kernel_ms_avg'],4))"
}
run regions CUGRAPH_AMD_MG_REGIONS=1
run oneregion CUGRAPH_AMD_MG_REGIONS=0
