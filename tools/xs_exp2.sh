#!/bin/bash
B="python3 -u bench.py --scale 26 --steps 5 --warmup 1 --no-cpu-baseline --no-convergence"
run() { tag=$1; shift; env "$@" timeout -k 10 200 $B > gpurun_out/x_$tag.json 2>/dev/null || echo fail $tag; }
run all A=1
run xsonly CUGRAPH_AMD_BUCKET_MASK=32
run xsonly_m8 CUGRAPH_AMD_BUCKET_MASK=32 CUGRAPH_AMD_GATHER_MASK=0xFF
CUGRAPH_AMD_BUCKET_MASK=32 timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_x3 -- $B > /dev/null 2>&1
echo xs-exp-ok
