#!/bin/bash
# calibration of FETCH_SIZE on this kernel's own streams (MI355X_MICROARCH.md: widths other than
# 16 B/lane are uncalibrated): (a) isolated-vertex bucket alone = 16 B/lane streaming reads of a
# known size; (b) XS tiles alone with every gather folded into 1 KiB = the 4 B/lane entry stream
# the masks / profile stamps exist only in the timing-aid build (make -C cugraph_amd/csrc AIDS=1)
export CUGRAPH_AMD_LIBRARY=${CUGRAPH_AMD_LIBRARY:-/root/repo/cugraph_amd/libcugraph_amd_c_aids.so}
B="bench.py --scale 26 --steps 5 --warmup 2 --no-cpu-baseline --no-convergence"
export TMPDIR=/tmp
CUGRAPH_AMD_BUCKET_MASK=16 timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_cal_iso -- python3 $B > /dev/null 2>&1 || echo a-fail
CUGRAPH_AMD_BUCKET_MASK=32 CUGRAPH_AMD_GATHER_MASK=0xFF timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_cal_xs -- python3 $B > /dev/null 2>&1 || echo b-fail
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 $B > gpurun_out/prof_stats.json 2>/dev/null || { echo stats-fail; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 $B > /dev/null 2>&1 || { echo fetch-fail; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 $B > /dev/null 2>&1 || { echo write-fail; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d gpurun_out/pmc_tcc -- python3 $B > /dev/null 2>&1 || { echo tcc-fail; exit 1; }
echo pmc-ok
