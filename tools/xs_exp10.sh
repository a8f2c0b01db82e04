#!/bin/bash
run() { name=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py --force-mg --steps 10 --warmup 3 --no-cpu-baseline --no-convergence > gpurun_out/t.json 2> gpurun_out/t.err || { echo fail $name; tail -3 gpurun_out/t.err; exit 1; }
  python3 -c "import json;d=json.load(open('gpurun_out/t.json'));print('$name', 'GTEPS', round(d['value'],1), 'ms', round(d['ms_per_step'],4), 'kern', round(d['roofline']['kernel_ms_avg'],4))"
}
run mg13 A=1
run mg14 CUGRAPH_AMD_XS_ROWBITS=14
run mg14p CUGRAPH_AMD_XS_ROWBITS=14 CUGRAPH_AMD_XS_PIECE=524288
