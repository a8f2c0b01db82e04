#!/bin/bash
# r3: where xs_kernel's time goes -- ablations under the timing-aid build (results of ablated runs are wrong)
# usage: tools/r3_ablate.sh <tag> [scale]
set -o pipefail
tag=$1; scale=${2:-26}
AIDS=$PWD/cugraph_amd/libcugraph_amd_c_aids.so
B="python3 -u bench.py --scale $scale --steps 10 --warmup 2 --no-cpu-baseline --no-convergence --no-extra-configs"
run() { # name, env...
  local name=$1; shift
  env "$@" timeout -k 10 200 $B > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err || return 1
  python3 -c "
import json,sys
d=json.load(open('gpurun_out/${tag}_$name.json'))
print('%-28s ms/step %.4f  xs_kernel %.4f'%('$name', d['ms_per_step'], d['roofline']['kernel_ms_avg']))"
}
run shipped_narrow14 X=1 &&
run shipped_narrow13 CUGRAPH_AMD_XS_ROWBITS=13 &&
run shipped_wide13 CUGRAPH_AMD_XS_WIDE=1 &&
run aids_base CUGRAPH_AMD_LIBRARY=$AIDS &&
run aids_nogather CUGRAPH_AMD_LIBRARY=$AIDS CUGRAPH_AMD_XS_NO_GATHER=1 &&
run aids_nogather_noatom CUGRAPH_AMD_LIBRARY=$AIDS CUGRAPH_AMD_XS_NO_GATHER=1 CUGRAPH_AMD_XS_NO_ATOMICS=1 &&
run aids_nogather_noatom_nowb CUGRAPH_AMD_LIBRARY=$AIDS CUGRAPH_AMD_XS_NO_GATHER=1 CUGRAPH_AMD_XS_NO_ATOMICS=1 CUGRAPH_AMD_XS_NO_WRITEBACK=1 &&
run aids_noatom CUGRAPH_AMD_LIBRARY=$AIDS CUGRAPH_AMD_XS_NO_ATOMICS=1 &&
run aids_l1 CUGRAPH_AMD_LIBRARY=$AIDS CUGRAPH_AMD_GATHER_MASK=0xFF &&
run aids_l2 CUGRAPH_AMD_LIBRARY=$AIDS CUGRAPH_AMD_GATHER_MASK=0xFFFFF &&
run aids_wide_base CUGRAPH_AMD_LIBRARY=$AIDS CUGRAPH_AMD_XS_WIDE=1 &&
run aids_wide_nogather_noatom CUGRAPH_AMD_LIBRARY=$AIDS CUGRAPH_AMD_XS_WIDE=1 CUGRAPH_AMD_XS_NO_GATHER=1 CUGRAPH_AMD_XS_NO_ATOMICS=1 &&
echo ablate-ok
