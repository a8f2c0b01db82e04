#!/bin/bash
# r3: the tiles of the low-degree rows alone (scale 26, tiles 600+): time of the shipped build, phases of the phase build
B="python3 -u bench.py --scale 26 --steps 10 --warmup 2 --no-cpu-baseline --no-convergence --no-extra-configs"
for r in 260:600 600:100000; do
  env "$@" CUGRAPH_AMD_XS_ONLY_TILES=$r timeout -k 10 200 $B > gpurun_out/t2.json 2> gpurun_out/t2.err || { tail -5 gpurun_out/t2.err; exit 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/t2.json'))
print('tiles $r  xs2_kernel %.4f ms'%(d['roofline']['kernel_ms_avg']))"
  env "$@" CUGRAPH_AMD_XS_ONLY_TILES=$r CUGRAPH_AMD_LIBRARY=$PWD/cugraph_amd/libcugraph_amd_c_phase.so timeout -k 10 200 $B > gpurun_out/t2p.json 2> gpurun_out/t2p.err || { tail -5 gpurun_out/t2p.err; exit 1; }
  grep -h "XS2PHASE" gpurun_out/t2p.err | tail -1
done
