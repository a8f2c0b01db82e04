#!/bin/bash
# r3: per-wave phase profile (make VARIANT=phase...) under env settings.  usage: tools/r3_phase.sh <tag> <scale> <lib-variant> [ENV=VAL ...]
tag=$1; scale=$2; v=$3; shift 3
env "$@" CUGRAPH_AMD_LIBRARY=$PWD/cugraph_amd/libcugraph_amd_c_$v.so timeout -k 10 200 python3 -u bench.py --scale $scale --steps 10 --warmup 2 --no-cpu-baseline --no-convergence --no-extra-configs > gpurun_out/${tag}.json 2> gpurun_out/${tag}.err || { echo "$tag failed"; tail -5 gpurun_out/${tag}.err; exit 1; }
python3 -c "
import json
d=json.load(open('gpurun_out/${tag}.json'))
print('%-28s ms/step %.4f  xs_kernel %.4f'%('$tag', d['ms_per_step'], d['roofline']['kernel_ms_avg']))"
grep -h "XS2\?PHASE" gpurun_out/${tag}.err | tail -1
