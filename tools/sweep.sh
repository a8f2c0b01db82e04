#!/bin/bash
# usage: tools/sweep.sh <tag> [scale]   -- per-bucket timing of the SpMV (timing aid; results of masked runs are wrong)
# the masks / profile stamps exist only in the timing-aid build (make -C cugraph_amd/csrc AIDS=1)
export CUGRAPH_AMD_LIBRARY=${CUGRAPH_AMD_LIBRARY:-/root/repo/cugraph_amd/libcugraph_amd_c_aids.so}
tag=$1; scale=${2:-26}
B="python3 -u bench.py --scale $scale --steps 5 --warmup 1 --no-cpu-baseline --no-convergence"
timeout -k 10 200 $B > gpurun_out/${tag}_base.json 2>gpurun_out/${tag}_base.err || exit 1
CUGRAPH_AMD_GATHER_MASK=0xFFFF timeout -k 10 200 $B > gpurun_out/${tag}_m16.json 2>/dev/null || exit 1
for m in 1 2 4 8 16; do
  CUGRAPH_AMD_BUCKET_MASK=$m timeout -k 10 200 $B > gpurun_out/${tag}_b$m.json 2>/dev/null || exit 1
  CUGRAPH_AMD_GATHER_MASK=0xFFFF CUGRAPH_AMD_BUCKET_MASK=$m timeout -k 10 200 $B > gpurun_out/${tag}_m16_b$m.json 2>/dev/null || exit 1
done
echo sweep-ok
