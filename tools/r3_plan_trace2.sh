#!/bin/bash
# r3: sort kernels of the plan build against sub-unit bits.  usage: tools/r3_plan_trace2.sh <scale> <tile_edges> <subbits> ...
scale=$1; te=$2; shift 2
for sb in "$@"; do
  rm -rf gpurun_out/ps_$sb
  CUGRAPH_AMD_XS_SORT_SUBBITS=$sb CUGRAPH_AMD_XS_TILE_EDGES=$te timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ps_$sb -- python3 bench.py --scale $scale --steps 10 --warmup 2 --no-cpu-baseline --no-convergence --no-extra-configs > gpurun_out/ps_$sb.json 2> gpurun_out/ps_$sb.err || { tail -5 gpurun_out/ps_$sb.err; exit 1; }
  python3 - <<PY
import json, csv, glob
d=json.load(open('gpurun_out/ps_$sb.json'))
rows=list(csv.DictReader(open(glob.glob('gpurun_out/ps_$sb/**/*kernel_stats.csv', recursive=True)[0])))
tot={}
for r in rows:
    n=r['Name'].replace('void cga::(anonymous namespace)::','').replace('cga::(anonymous namespace)::','').split('(')[0].split('<')[0]
    tot[n]=tot.get(n,0)+float(r['TotalDurationNs'])/1e6
print('scale $scale tile_edges $te subbits $sb: plan %.3f s, ms/step %.4f xs2 %.4f | sort_big %.1f sort_medium %.1f sort_wave %.1f sort_small %.1f scatter_lds %.1f hist_lds %.1f ms'%(
  d['graph_build']['plan_build_s'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], tot.get('sort_big_kernel',0), tot.get('sort_medium_kernel',0), tot.get('sort_wave_kernel',0), tot.get('sort_small_kernel',0), tot.get('scatter_lds_kernel',0), tot.get('hist_lds_kernel',0)))
PY
done
