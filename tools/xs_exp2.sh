#!/bin/bash
B="python3 -u bench.py --scale 26 --steps 5 --warmup 1 --no-cpu-baseline --no-convergence"
run() { tag=$1; shift; env "$@" timeout -k 10 200 $B > gpurun_out/x_$tag.json 2>/dev/null || echo fail $tag; }
run all A=1
run dedupe CUGRAPH_AMD_XS_DEDUPE=1
run dedupe_m8 CUGRAPH_AMD_XS_DEDUPE=1 CUGRAPH_AMD_BUCKET_MASK=32 CUGRAPH_AMD_GATHER_MASK=0xFF
timeout -k 10 300 $B --force-mg > gpurun_out/x_mg.json 2>gpurun_out/x_mg.err || echo fail mg
echo xs-exp-ok
