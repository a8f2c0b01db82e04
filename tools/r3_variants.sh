#!/bin/bash
# r3: time library variants (make VARIANT=...) against the shipped build.  usage: tools/r3_variants.sh <tag> <scale> <variant>...
set -o pipefail
tag=$1; scale=$2; shift 2
B="python3 -u bench.py --scale $scale --steps 10 --warmup 2 --no-cpu-baseline --no-convergence --no-extra-configs"
run() { # name, env...
  local name=$1; shift
  env "$@" timeout -k 10 200 $B > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err || { echo "$name failed"; tail -5 gpurun_out/${tag}_$name.err; return 1; }
  python3 -c "
import json,sys
d=json.load(open('gpurun_out/${tag}_$name.json'))
print('%-28s ms/step %.4f  xs_kernel %.4f  frac %.3f'%('$name', d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac']))"
}
for v in "$@"; do
  case $v in
    shipped) run shipped X=1 || exit 1;;
    wide) run wide CUGRAPH_AMD_XS_WIDE=1 || exit 1;;
    *) run $v CUGRAPH_AMD_LIBRARY=$PWD/cugraph_amd/libcugraph_amd_c_$v.so || exit 1;;
  esac
done
echo variants-ok
