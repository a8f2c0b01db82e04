#!/bin/bash
# round-2 profile set of the shipped build: kernel-trace stats, then one PMC pass per counter set (program directly
# after `--`), plus the two calibration runs of FETCH_SIZE on this kernel's own streams with the timing-aid build.
B="bench.py --scale 26 --steps 5 --warmup 2 --no-cpu-baseline --no-convergence --no-extra-configs"
AIDS=/root/repo/cugraph_amd/libcugraph_amd_c_aids.so
export TMPDIR=/tmp
rm -rf gpurun_out/r2_stats gpurun_out/r2_fetch gpurun_out/r2_write gpurun_out/r2_tcc gpurun_out/r2_cal_xs gpurun_out/r2_sq
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_stats -- python3 $B > gpurun_out/r2_stats.json 2>/dev/null || { echo stats-fail; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r2_fetch -- python3 $B > /dev/null 2>&1 || { echo fetch-fail; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r2_write -- python3 $B > /dev/null 2>&1 || { echo write-fail; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d gpurun_out/r2_tcc -- python3 $B > /dev/null 2>&1 || { echo tcc-fail; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d gpurun_out/r2_sq -- python3 $B > /dev/null 2>&1 || { echo sq-fail; exit 1; }
CUGRAPH_AMD_LIBRARY=$AIDS CUGRAPH_AMD_BUCKET_MASK=32 CUGRAPH_AMD_GATHER_MASK=0xFF timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r2_cal_xs -- python3 $B > /dev/null 2>&1 || { echo cal-fail; exit 1; }
python3 - <<'PY'
import csv, glob, json, collections
def per_kernel(d, match):
    f = glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        if match in r["Kernel_Name"]:
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {c: sum(v.values()) / len(v) for c, v in acc.items()}, (len(next(iter(acc.values()))) if acc else 0)
out = {}
for d in ("r2_fetch", "r2_write", "r2_tcc", "r2_sq", "r2_cal_xs"):
    out[d], out[d + "_launches"] = per_kernel(d, "xs_kernel")
ks = {}
for r in csv.DictReader(open(glob.glob("gpurun_out/r2_stats/**/*kernel_stats.csv", recursive=True)[0])):
    for k in ("xs_kernel", "spmv_kernel", "xs_finish_kernel", "finalize_kernel"):
        if k + "<" in r["Name"] or k + "(" in r["Name"]:
            ks[k] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6}
out["kernel_stats"] = ks
json.dump(out, open("gpurun_out/r2_pmc_raw.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
cp $(find gpurun_out/r2_stats -name '*kernel_stats.csv' | head -1) gpurun_out/r2_final_scale26_kernel_stats.csv
echo profile-ok
