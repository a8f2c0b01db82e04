#!/bin/bash
# round-3 profile set of the shipped build (program directly after `--`, one counter set per run):
#   kernel-trace stats of the bench command, FETCH_SIZE / WRITE_SIZE / TCC passes for the headline config (scale 26),
#   the FETCH_SIZE calibration on the kernel's own entry stream (timing-aid build, gathers folded into 1 KiB),
#   FETCH_SIZE / WRITE_SIZE passes for C2 (scale 22, natural ids) and C3 (scale 24 fp64 weighted personalized).
B26="bench.py --scale 26 --steps 5 --warmup 2 --no-cpu-baseline --no-convergence --no-extra-configs"
B22="bench.py --scale 22 --scramble 0 --steps 5 --warmup 2 --no-cpu-baseline --no-convergence --no-extra-configs"
C3="tools/c3_bench.py 24"
AIDS=$PWD/cugraph_amd/libcugraph_amd_c_aids.so
export TMPDIR=/tmp
pass() { # dir, counters (quoted), program...
  local d=$1 c=$2; shift 2
  rm -rf gpurun_out/$d
  timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/$d -- python3 "$@" > /dev/null 2> gpurun_out/$d.err || { echo "$d-fail"; tail -3 gpurun_out/$d.err; exit 1; }
}
rm -rf gpurun_out/r3_stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_stats -- python3 $B26 > gpurun_out/r3_stats.json 2>/dev/null || { echo stats-fail; exit 1; }
pass r3_fetch "FETCH_SIZE" $B26
pass r3_write "WRITE_SIZE" $B26
pass r3_tcc "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" $B26
pass r3_sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" $B26
CUGRAPH_AMD_LIBRARY=$AIDS CUGRAPH_AMD_BUCKET_MASK=32 CUGRAPH_AMD_GATHER_MASK=0xFF pass r3_cal_xs "FETCH_SIZE" $B26
pass r3_c2_fetch "FETCH_SIZE" $B22
pass r3_c2_write "WRITE_SIZE" $B22
pass r3_c3_fetch "FETCH_SIZE" $C3
pass r3_c3_write "WRITE_SIZE" $C3
rm -rf gpurun_out/r3_c3_stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_c3_stats -- python3 $C3 > gpurun_out/r3_c3_stats.txt 2>/dev/null || { echo c3-stats-fail; exit 1; }
rm -rf gpurun_out/r3_c2_stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_c2_stats -- python3 $B22 > gpurun_out/r3_c2_stats.json 2>/dev/null || { echo c2-stats-fail; exit 1; }
python3 - <<'PY'
import csv, glob, json, collections
def per_kernel(d, match):
    f = glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    names = set()
    for r in csv.DictReader(open(f[0])):
        if match in r["Kernel_Name"]:
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
            names.add(r["Kernel_Name"].split("(")[0])
    out = {c: sum(v.values()) / len(v) for c, v in acc.items()}
    out["_launches"] = len(next(iter(acc.values()))) if acc else 0
    out["_kernel"] = sorted(names)
    return out
def stats(d):
    ks = {}
    for r in csv.DictReader(open(glob.glob(f"gpurun_out/{d}/**/*kernel_stats.csv", recursive=True)[0])):
        for k in ("xs2_kernel", "xs_kernel", "spmv_kernel", "xs_finish_kernel", "finalize_kernel", "pers_fix_kernel"):
            if k + "<" in r["Name"] or k + "(" in r["Name"]:
                ks[k] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6, "name": r["Name"].split("(")[0]}
    return ks
out = {}
for d in ("r3_fetch", "r3_write", "r3_tcc", "r3_sq", "r3_cal_xs", "r3_c2_fetch", "r3_c2_write", "r3_c3_fetch", "r3_c3_write"):
    out[d] = per_kernel(d, "xs")  # xs_kernel / xs2_kernel (not xs_finish: matched apart below)
    out[d + "_finish"] = per_kernel(d, "xs_finish_kernel")
for d in ("r3_stats", "r3_c2_stats", "r3_c3_stats"):
    out[d] = stats(d)
json.dump(out, open("gpurun_out/r3_pmc_raw.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if "finish" not in k}, indent=1)[:6000])
PY
cp $(find gpurun_out/r3_stats -name '*kernel_stats.csv' | head -1) gpurun_out/r3_final_scale26_kernel_stats.csv
cp $(find gpurun_out/r3_c2_stats -name '*kernel_stats.csv' | head -1) gpurun_out/r3_c2_scale22_kernel_stats.csv
cp $(find gpurun_out/r3_c3_stats -name '*kernel_stats.csv' | head -1) gpurun_out/r3_c3_scale24_kernel_stats.csv
echo profile-ok
