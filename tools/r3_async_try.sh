#!/bin/bash
# r3: xs2_kernel (CUGRAPH_AMD_XS_ASYNC=1): timing-aid build first (bounds-checked), then the shipped build.
set -o pipefail
mkdir -p gpurun_out
export CUGRAPH_AMD_XS_ASYNC=1
AIDS=$PWD/cugraph_amd/libcugraph_amd_c_aids.so
B="python3 -u bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-convergence --no-extra-configs"
CUGRAPH_AMD_LIBRARY=$AIDS timeout -k 10 300 python3 -u -m pytest tests/test_gpu_parity_small.py -m gpu -x -q --timeout 60 --timeout-method thread > gpurun_out/r3_async_aids_tests.txt 2>&1 || { tail -30 gpurun_out/r3_async_aids_tests.txt; exit 1; }
echo aids-tests-ok; tail -2 gpurun_out/r3_async_aids_tests.txt
CUGRAPH_AMD_LIBRARY=$AIDS timeout -k 10 120 $B --scale 22 > gpurun_out/r3_async_aids22.json 2> gpurun_out/r3_async_aids22.err || { tail -5 gpurun_out/r3_async_aids22.err; exit 1; }
grep "bounds check" gpurun_out/r3_async_aids22.err
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_parity_small.py tests/test_gpu_edge_cases.py -m gpu -x -q --timeout 60 --timeout-method thread > gpurun_out/r3_async_tests.txt 2>&1 || { tail -30 gpurun_out/r3_async_tests.txt; exit 1; }
echo tests-ok; tail -2 gpurun_out/r3_async_tests.txt
for sc in 22 24 26; do
  timeout -k 10 200 $B --scale $sc > gpurun_out/r3_async$sc.json 2> gpurun_out/r3_async$sc.err || { tail -5 gpurun_out/r3_async$sc.err; exit 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/r3_async$sc.json'))
print('async scale $sc ms/step %.4f  xs_kernel %.4f  frac %.3f'%(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac']))"
done
