#!/bin/bash
B="python3 -u bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-convergence --no-extra-configs"
for sc in 24 26; do for bs in 0 1 2; do
  CUGRAPH_AMD_XS_ASYNC=1 CUGRAPH_AMD_SPMV_BESIDE=$bs timeout -k 10 200 $B --scale $sc > gpurun_out/r3bs.json 2> gpurun_out/r3bs.err || { tail -5 gpurun_out/r3bs.err; exit 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/r3bs.json'))
print('scale $sc beside=$bs ms/step %.4f  xs_kernel %.4f'%(d['ms_per_step'], d['roofline']['kernel_ms_avg']))"
done; done
