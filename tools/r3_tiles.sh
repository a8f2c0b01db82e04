#!/bin/bash
# r3: xs_kernel time by tile class (CUGRAPH_AMD_XS_ONLY_TILES: timing aid, results wrong).  usage: tools/r3_tiles.sh <tag> <scale> [ENV=VAL...]
tag=$1; scale=$2; shift 2
B="python3 -u bench.py --scale $scale --steps 10 --warmup 2 --no-cpu-baseline --no-convergence --no-extra-configs"
for r in 0:100000 0:20 20:60 60:130 130:300 300:100000 0:130 130:100000; do
  env "$@" CUGRAPH_AMD_XS_ONLY_TILES=$r timeout -k 10 200 $B > gpurun_out/${tag}_t.json 2> gpurun_out/${tag}_t.err || { tail -5 gpurun_out/${tag}_t.err; exit 1; }
  [ $r = 0:100000 ] && grep "xs tile" gpurun_out/${tag}_t.err | head -40
  grep "xs only" gpurun_out/${tag}_t.err | head -1
  python3 -c "
import json
d=json.load(open('gpurun_out/${tag}_t.json'))
print('tiles $r  xs_kernel %.4f ms'%(d['roofline']['kernel_ms_avg']))"
done
