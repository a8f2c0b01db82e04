#!/bin/bash
# usage: tools/r2_pmc.sh name [ENV=val ...] -- fabric / L2 request counters of xs_kernel for one setting
name=$1; shift
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp
B="bench.py --scale ${SCALE:-26} --steps 5 --warmup 2 --no-cpu-baseline --no-convergence --no-extra-configs"
rm -rf gpurun_out/pmc_$name
timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d gpurun_out/pmc_$name -- python3 $B > /dev/null 2>&1 || { echo "$name pmc-fail"; exit 1; }
python3 - "$name" <<'PY'
import csv, glob, sys, collections
name = sys.argv[1]
f = glob.glob(f"gpurun_out/pmc_{name}/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "xs_kernel" in k or "xs_finish" in k or "spmv_kernel" in k:
        key = "xs_kernel" if "xs_kernel" in k else "xs_finish" if "xs_finish" in k else "spmv_kernel"
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "TCC_HIT_sum": n[key] += 1
for key in acc:
    print(name, key, "launches", n[key], {c: round(v / max(n[key], 1) / 1e6, 2) for c, v in acc[key].items()}, "(M per launch)")
PY
