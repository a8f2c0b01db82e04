#!/bin/bash
B="python3 -u bench.py --scale 26 --steps 5 --warmup 1 --no-cpu-baseline --no-convergence"
run() { tag=$1; shift; env "$@" timeout -k 10 200 $B > gpurun_out/x_$tag.json 2>gpurun_out/x_$tag.err || echo fail $tag; }
run base A=1
run b768 CUGRAPH_AMD_XS_BLOCKS=768
run b1280 CUGRAPH_AMD_XS_BLOCKS=1280
echo xs-exp-ok
