#!/bin/bash
# usage: kstats.sh name [ENV=val ...]   -> per-kernel average durations of a short bench run
name=$1; shift
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rm -rf gpurun_out/ks_$name
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$name -- python3 bench.py --scale ${SCALE:-26} --steps 5 --warmup 2 --no-cpu-baseline --no-convergence > gpurun_out/ks_$name.json 2> gpurun_out/ks_$name.err || { echo fail; tail -5 gpurun_out/ks_$name.err; exit 1; }
f=$(find gpurun_out/ks_$name -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("xs_kernel", "spmv_kernel", "finish", "finalize", "pers_fix")):
        print(f"{n[:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e6:8.4f} ms")
PY
