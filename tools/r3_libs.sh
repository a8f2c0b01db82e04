#!/bin/bash
# r3: the same bench line under several builds of the library.  usage: tools/r3_libs.sh <scale> <variant> [<variant> ...]  ("-" = shipped)
scale=$1; shift
B="python3 -u bench.py --scale $scale --steps 10 --warmup 2 --no-cpu-baseline --no-convergence --no-extra-configs"
for v in "$@"; do
  lib=$PWD/cugraph_amd/libcugraph_amd_c_$v.so; [ "$v" = "-" ] && lib=$PWD/cugraph_amd/libcugraph_amd_c.so
  CUGRAPH_AMD_LIBRARY=$lib timeout -k 10 200 $B > gpurun_out/lib_$v.json 2> gpurun_out/lib_$v.err || { echo "$v failed"; tail -5 gpurun_out/lib_$v.err; exit 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/lib_$v.json'))
print('scale $scale lib %-8s ms/step %.4f dominant kernel %.4f ms'%('$v', d['ms_per_step'], d['roofline']['kernel_ms_avg']))"
done
