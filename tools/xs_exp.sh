#!/bin/bash
B="python3 -u bench.py --scale 26 --steps 5 --warmup 1 --no-cpu-baseline --no-convergence"
run() { tag=$1; shift; env "$@" timeout -k 10 200 $B > gpurun_out/x_$tag.json 2>/dev/null || echo fail $tag; }
run xsonly CUGRAPH_AMD_BUCKET_MASK=32
run xsonly_m16 CUGRAPH_AMD_BUCKET_MASK=32 CUGRAPH_AMD_GATHER_MASK=0xFFFF
run aff1 CUGRAPH_AMD_BUCKET_MASK=32 CUGRAPH_AMD_XS_AFFINITY=1
run aff2 CUGRAPH_AMD_BUCKET_MASK=32 CUGRAPH_AMD_XS_AFFINITY=2
run blk512 CUGRAPH_AMD_BUCKET_MASK=32 CUGRAPH_AMD_XS_BLOCKS=512
run blk1280 CUGRAPH_AMD_BUCKET_MASK=32 CUGRAPH_AMD_XS_BLOCKS=1280
run blk2048 CUGRAPH_AMD_BUCKET_MASK=32 CUGRAPH_AMD_XS_BLOCKS=2048
CUGRAPH_AMD_BUCKET_MASK=32 timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_x1 -- $B > /dev/null 2>&1
CUGRAPH_AMD_BUCKET_MASK=32 CUGRAPH_AMD_XS_AFFINITY=2 timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_x2 -- $B > /dev/null 2>&1
echo xs-exp-ok
