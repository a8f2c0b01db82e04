#!/bin/bash
# r3: bench line + per-kernel split under several builds.  usage: tools/r3_libs2.sh <scale> <variant> ...   ("-" = shipped)
scale=$1; shift
for v in "$@"; do
  lib=$PWD/cugraph_amd/libcugraph_amd_c_$v.so; [ "$v" = "-" ] && lib=$PWD/cugraph_amd/libcugraph_amd_c.so
  rm -rf gpurun_out/ks_$v
  CUGRAPH_AMD_LIBRARY=$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$v -- python3 bench.py --scale $scale --steps 10 --warmup 2 --no-cpu-baseline --no-convergence --no-extra-configs > gpurun_out/lib_$v.json 2> gpurun_out/lib_$v.err || { echo "$v failed"; tail -5 gpurun_out/lib_$v.err; exit 1; }
  python3 - <<PY
import json, csv, glob
d=json.load(open('gpurun_out/lib_$v.json'))
print('scale $scale lib %-6s ms/step %.4f dominant %.4f ms  build %.3f + plan %.3f s'%('$v', d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['graph_build']['coo_to_csc_s'], d['graph_build']['plan_build_s']), end='  |')
for r in csv.DictReader(open(glob.glob('gpurun_out/ks_$v/**/*kernel_stats.csv', recursive=True)[0])):
    for k in ('xs2_kernel<','spmv_kernel<','xs_finish_kernel','finalize_kernel'):
        if '::'+k in r['Name']: print(' %s %.4f'%(k.rstrip('<'), float(r['AverageNs'])/1e6), end='')
print()
PY
done
