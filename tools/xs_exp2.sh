#!/bin/bash
B="python3 -u bench.py --scale 26 --steps 5 --warmup 1 --no-cpu-baseline --no-convergence"
run() { tag=$1; shift; env "$@" timeout -k 10 200 $B > gpurun_out/x_$tag.json 2>gpurun_out/x_$tag.err || echo fail $tag; }
run all A=1
run nodedupe CUGRAPH_AMD_XS_DEDUPE=0
run m20 CUGRAPH_AMD_GATHER_MASK=0xFFFFF
run m8 CUGRAPH_AMD_GATHER_MASK=0xFF
echo xs-exp-ok
