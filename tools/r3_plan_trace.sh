#!/bin/bash
# r3: which kernels of the plan build grow with the entries per tile.  usage: tools/r3_plan_trace.sh <scale> <tile_edges> ...
scale=$1; shift
for te in "$@"; do
  rm -rf gpurun_out/pt_$te
  CUGRAPH_AMD_XS_TILE_EDGES=$te timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pt_$te -- python3 bench.py --scale $scale --steps 3 --warmup 1 --no-cpu-baseline --no-convergence --no-extra-configs > gpurun_out/pt_$te.json 2> gpurun_out/pt_$te.err || { tail -5 gpurun_out/pt_$te.err; exit 1; }
  python3 - <<PY
import json, csv, glob
d=json.load(open('gpurun_out/pt_$te.json'))
print('scale $scale tile_edges $te: build %.3f s plan %.3f s, ms/step %.3f'%(d['graph_build']['coo_to_csc_s'], d['graph_build']['plan_build_s'], d['ms_per_step']))
rows=list(csv.DictReader(open(glob.glob('gpurun_out/pt_$te/**/*kernel_stats.csv', recursive=True)[0])))
for r in rows[:14]:
    n=r['Name'].replace('void cga::(anonymous namespace)::','').replace('cga::(anonymous namespace)::','').split('(')[0][:60]
    print('   %-62s calls %4s total %9.2f ms'%(n, r['Calls'], float(r['TotalDurationNs'])/1e6))
PY
done
