#!/bin/bash
# r3: xs_kernel time of a tile class under the compile-time ablations.  usage: tools/r3_tiles_abl.sh <tag> <scale> <lo:hi> [ENV=VAL...]
tag=$1; scale=$2; r=$3; shift 3
B="python3 -u bench.py --scale $scale --steps 10 --warmup 2 --no-cpu-baseline --no-convergence --no-extra-configs"
for v in nodense nog l1 noatom nognoatom; do
  env "$@" CUGRAPH_AMD_LIBRARY=$PWD/cugraph_amd/libcugraph_amd_c_$v.so CUGRAPH_AMD_XS_ONLY_TILES=$r timeout -k 10 200 $B > gpurun_out/${tag}_t.json 2> gpurun_out/${tag}_t.err || { tail -5 gpurun_out/${tag}_t.err; exit 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/${tag}_t.json'))
print('tiles $r  %-10s xs_kernel %.4f ms'%('$v', d['roofline']['kernel_ms_avg']))"
done
