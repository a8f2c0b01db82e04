#!/bin/bash
# ablation of xs_kernel: 1 = no LDS atomics, 2 = fp32 LDS atomics, 3 = no gathers, 4 = stream only
B="bench.py --scale 26 --steps 10 --warmup 2 --no-cpu-baseline --no-convergence"
for m in 0 1 2 3 4; do
  CUGRAPH_AMD_XS_EXP=$m timeout -k 10 200 python3 $B > gpurun_out/exp_$m.json 2> gpurun_out/exp_$m.err || { echo fail $m; exit 1; }
  python3 -c "import json;d=json.load(open('gpurun_out/exp_$m.json'));print($m, d['ms_per_step'], d['roofline']['kernel_ms_avg'])"
done
