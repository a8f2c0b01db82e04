#!/bin/bash
B="python3 -u bench.py --scale 26 --steps 5 --warmup 1 --no-cpu-baseline --no-convergence"
run() { tag=$1; shift; env "$@" timeout -k 10 200 $B > gpurun_out/x_$tag.json 2>gpurun_out/x_$tag.err || echo fail $tag; }
run base A=1
run b8 CUGRAPH_AMD_BUCKET_MASK=8
run b16 CUGRAPH_AMD_BUCKET_MASK=16
run xsonly CUGRAPH_AMD_BUCKET_MASK=32
echo xs-exp-ok
