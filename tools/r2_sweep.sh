#!/bin/bash
# usage: [NOCHECK=1] [SCALE=26] tools/r2_sweep.sh name:ENV=val[,ENV=val] ...
# per named setting: a correctness check at scale 20 against the oracle (unless NOCHECK: timing-aid
# settings give wrong results on purpose), then one short bench run (ms per iteration, xs_kernel ms)
B="bench.py --scale ${SCALE:-26} --steps 10 --warmup 2 --no-cpu-baseline --no-convergence"
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}; [ "$envs" = "$spec" ] && envs="A=1"
  if [ -n "$NOCHECK" ]; then
    echo "unchecked" > gpurun_out/chk_$name.txt
  elif ! env ${envs//,/ } timeout -k 10 120 python3 tools/variant_check.py 20 > gpurun_out/chk_$name.txt 2> gpurun_out/chk_$name.err; then
    echo "CHECK FAIL $name"; tail -3 gpurun_out/chk_$name.err; tail -1 gpurun_out/chk_$name.txt; continue
  fi
  env ${envs//,/ } timeout -k 10 200 python3 $B > gpurun_out/sw_$name.json 2> gpurun_out/sw_$name.err || { echo "fail $name"; tail -3 gpurun_out/sw_$name.err; exit 1; }
  python3 -c "import json;d=json.load(open('gpurun_out/sw_$name.json'));print('$name', 'iter_ms', round(d['ms_per_step'],3), 'xs_ms', round(d['roofline']['kernel_ms_avg'],3), 'frac', round(d['roofline']['frac'],3), open('gpurun_out/chk_$name.txt').read().strip()[-110:])"
done
