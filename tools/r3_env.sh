#!/bin/bash
# r3: bench line + per-kernel split under env settings.  usage: tools/r3_env.sh <scale> "ENV=VAL ENV=VAL" ...   ("-" = none)
scale=$1; shift
i=0
for e in "$@"; do
  i=$((i+1)); [ "$e" = "-" ] && e="CGA_NONE=1"
  rm -rf gpurun_out/ke_$i
  env $e timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ke_$i -- python3 bench.py --scale $scale --steps 10 --warmup 2 --no-cpu-baseline --no-convergence --no-extra-configs > gpurun_out/ke_$i.json 2> gpurun_out/ke_$i.err || { echo "$e failed"; tail -5 gpurun_out/ke_$i.err; exit 1; }
  python3 - <<PY
import json, csv, glob
d=json.load(open('gpurun_out/ke_$i.json'))
print('scale $scale [%s] ms/step %.4f dominant %.4f ms  build %.3f + plan %.3f s'%('$e', d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['graph_build']['coo_to_csc_s'], d['graph_build']['plan_build_s']), end='  |')
for r in csv.DictReader(open(glob.glob('gpurun_out/ke_$i/**/*kernel_stats.csv', recursive=True)[0])):
    for k in ('xs2_kernel<','xs_kernel<','spmv_kernel<','xs_finish_kernel','finalize_kernel'):
        if '::'+k in r['Name']: print(' %s %.4f'%(k.rstrip('<'), float(r['AverageNs'])/1e6), end='')
print()
PY
done
