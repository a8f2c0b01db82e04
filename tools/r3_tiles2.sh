#!/bin/bash
# r3: xs2_kernel (8192-row tiles) by tile class: time of the shipped build and the phase profile of the phase build
B="python3 -u bench.py --scale 26 --steps 10 --warmup 2 --no-cpu-baseline --no-convergence --no-extra-configs"
for r in 0:100000 0:40 40:120 120:260 260:600 600:100000; do
  CUGRAPH_AMD_XS_ONLY_TILES=$r timeout -k 10 200 $B > gpurun_out/t2.json 2> gpurun_out/t2.err || { tail -5 gpurun_out/t2.err; exit 1; }
  [ $r = 0:100000 ] && grep "xs tile" gpurun_out/t2.err | head -40
  grep "xs only" gpurun_out/t2.err | head -1
  python3 -c "
import json
d=json.load(open('gpurun_out/t2.json'))
print('tiles $r  xs2_kernel %.4f ms'%(d['roofline']['kernel_ms_avg']))"
  CUGRAPH_AMD_XS_ONLY_TILES=$r CUGRAPH_AMD_LIBRARY=$PWD/cugraph_amd/libcugraph_amd_c_phase.so timeout -k 10 200 $B > gpurun_out/t2p.json 2> gpurun_out/t2p.err || { tail -5 gpurun_out/t2p.err; exit 1; }
  grep -h "XS2PHASE" gpurun_out/t2p.err | tail -1
done
