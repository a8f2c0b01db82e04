#!/bin/bash
# r3: quick check of a build: small parity tests (sync and async xs kernels), then timings.  usage: tools/r3_quick.sh <tag> [scales...]
set -o pipefail
tag=$1; shift
scales=${@:-"22 24 26"}
B="python3 -u bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-convergence --no-extra-configs"
for a in 0 1; do
  CUGRAPH_AMD_XS_ASYNC=$a timeout -k 10 300 python3 -u -m pytest tests/test_gpu_parity_small.py tests/test_gpu_edge_cases.py -m gpu -x -q --timeout 60 --timeout-method thread > gpurun_out/${tag}_tests$a.txt 2>&1 || { tail -30 gpurun_out/${tag}_tests$a.txt; exit 1; }
  echo "tests async=$a: $(tail -1 gpurun_out/${tag}_tests$a.txt)"
done
for sc in $scales; do for a in 0 1; do
  CUGRAPH_AMD_XS_ASYNC=$a timeout -k 10 200 $B --scale $sc > gpurun_out/${tag}_s${sc}_a$a.json 2> gpurun_out/${tag}_s${sc}_a$a.err || { tail -5 gpurun_out/${tag}_s${sc}_a$a.err; exit 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/${tag}_s${sc}_a$a.json'))
print('scale $sc async=$a ms/step %.4f  xs_kernel %.4f  frac %.3f'%(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac']))"
done; done
