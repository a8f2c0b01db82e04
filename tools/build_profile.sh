#!/bin/bash
# usage: build_profile.sh [scale]  -> per-kernel totals of the one-off steps (edge generation, COO->CSC with
# renumbering, pull-plan build) from a rocprofv3 kernel trace of tools/plan_time.py
scale=${1:-26}
export TMPDIR=/tmp
rm -rf gpurun_out/bp_$scale
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bp_$scale -- python3 tools/plan_time.py $scale > gpurun_out/bp_$scale.txt 2> gpurun_out/bp_$scale.err || { echo fail; tail -5 gpurun_out/bp_$scale.err; exit 1; }
cat gpurun_out/bp_$scale.txt
f=$(find gpurun_out/bp_$scale -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms")
for r in rows[:40]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} total {float(r['TotalDurationNs'])/1e6:9.3f} ms")
PY
