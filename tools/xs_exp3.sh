#!/bin/bash
B="python3 -u bench.py --scale 26 --steps 5 --warmup 1 --no-cpu-baseline --no-convergence"
run() { tag=$1; shift; env "$@" timeout -k 10 200 $B > gpurun_out/x_$tag.json 2>/dev/null || echo fail $tag; }
run te256k CUGRAPH_AMD_XS_TILE_EDGES=262144
run te4m CUGRAPH_AMD_XS_TILE_EDGES=4194304
run te16m CUGRAPH_AMD_XS_TILE_EDGES=16777216
run sh7 CUGRAPH_AMD_XS_SLICE_SHIFT=7
run sh10 CUGRAPH_AMD_XS_SLICE_SHIFT=10
run sh14 CUGRAPH_AMD_XS_SLICE_SHIFT=14
run blk768 CUGRAPH_AMD_XS_BLOCKS=768
run blk1280 CUGRAPH_AMD_XS_BLOCKS=1280
echo xs-exp-ok
