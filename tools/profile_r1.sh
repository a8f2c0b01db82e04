#!/bin/bash
# round-1 profile set: default bench line, kernel-trace stats, PMC passes (one counter set per run)
set -o pipefail
B="bench.py --scale 26 --steps 5 --warmup 2 --no-cpu-baseline --no-convergence"
export TMPDIR=/tmp
rm -rf gpurun_out/prof_stats gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_tcc
timeout -k 10 600 python3 -u bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || { echo bench-default-fail; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 $B > gpurun_out/prof_stats.json 2>/dev/null || { echo stats-fail; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 $B > /dev/null 2>&1 || { echo fetch-fail; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 $B > /dev/null 2>&1 || { echo write-fail; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d gpurun_out/pmc_tcc -- python3 $B > /dev/null 2>&1 || { echo tcc-fail; exit 1; }
echo profile-ok
