#!/bin/bash
# kernel trace of one scale-26 graph + plan build (the bench with 2 steps): which kernels the build spends its time in
export TMPDIR=/tmp
rm -rf gpurun_out/build_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/build_trace -- python3 bench.py --scale ${SCALE:-26} --steps 2 --warmup 1 --no-cpu-baseline --no-convergence --no-extra-configs > gpurun_out/build_trace.json 2>/dev/null || { echo trace-fail; exit 1; }
cp $(find gpurun_out/build_trace -name '*kernel_stats.csv' | head -1) gpurun_out/build_trace_stats.csv
cp $(find gpurun_out/build_trace -name '*kernel_trace.csv' | head -1) gpurun_out/build_trace_trace.csv
echo trace-ok
