#!/bin/bash
# per-workgroup set-up / pass / write-back times of xs_kernel; args: ENV=val ...
# the masks / profile stamps exist only in the timing-aid build (make -C cugraph_amd/csrc AIDS=1)
export CUGRAPH_AMD_LIBRARY=${CUGRAPH_AMD_LIBRARY:-/root/repo/cugraph_amd/libcugraph_amd_c_aids.so}
for kv in "$@"; do export "$kv"; done
CUGRAPH_AMD_XS_PROFILE=1 timeout -k 10 200 python3 bench.py --scale ${SCALE:-26} --steps 4 --warmup 0 --no-cpu-baseline --no-convergence > gpurun_out/p5.json 2> gpurun_out/p5.err || exit 1
python3 tools/xs_prof.py gpurun_out/p5.err > gpurun_out/p5.txt; sed -n 1,8p gpurun_out/p5.txt
