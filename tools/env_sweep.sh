#!/bin/bash
# usage: tools/env_sweep.sh name:ENV=val[,ENV=val] ...   -- one short scale-26 bench run per named setting
B="bench.py --scale 26 --steps 10 --warmup 2 --no-cpu-baseline --no-convergence"
run() { # name, env...
  n=$1; shift
  env "$@" timeout -k 10 200 python3 $B > gpurun_out/sw_$n.json 2> gpurun_out/sw_$n.err || { echo fail $n; tail -3 gpurun_out/sw_$n.err; exit 1; }
  python3 -c "import json;d=json.load(open('gpurun_out/sw_$n.json'));print('$n', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms_avg'],3))"
}
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  run $name ${envs//,/ }
done
