#!/bin/bash
# r3: xs2_kernel by tile class under the compile-time ablations (variants nog / noa / non: no gathers, no atomics, neither)
B="python3 -u bench.py --scale 26 --steps 10 --warmup 2 --no-cpu-baseline --no-convergence --no-extra-configs"
for v in "" _nog _noa _non; do
for r in 0:100000 0:40 40:120 120:260 260:600 600:100000; do
  CUGRAPH_AMD_LIBRARY=$PWD/cugraph_amd/libcugraph_amd_c$v.so CUGRAPH_AMD_XS_ONLY_TILES=$r timeout -k 10 200 $B > gpurun_out/t3.json 2> gpurun_out/t3.err || { tail -5 gpurun_out/t3.err; exit 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/t3.json'))
print('variant [$v] tiles $r  xs2_kernel %.4f ms'%(d['roofline']['kernel_ms_avg']))"
done; done
