"""Fold the rocprofv3 passes of tools/pmc_passes.sh (gpurun_out/) into profiles/<tag>_pmc_xs.{json,md} inputs:
per-launch averages of every counter for xs_kernel, plus the two calibration runs."""
import csv, glob, json, sys, collections

def per_kernel(dirname, match):
    """counter -> (sum over launches of the per-launch value) / launches, for kernels whose name contains match"""
    f = glob.glob(f"gpurun_out/{dirname}/*/*counter_collection.csv")
    if not f:
        return {}
    acc = collections.defaultdict(lambda: collections.defaultdict(float))  # counter -> dispatch -> value
    for r in csv.DictReader(open(f[0])):
        if match in r["Kernel_Name"]:
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {c: (sum(d.values()) / len(d), len(d)) for c, d in acc.items()}

out = {}
for d, m in (("pmc_fetch", "xs_kernel"), ("pmc_write", "xs_kernel"), ("pmc_tcc", "xs_kernel"),
             ("pmc_cal_xs", "xs_kernel"), ("pmc_cal_iso", "spmv_kernel")):
    out[d] = per_kernel(d, m)
stats = glob.glob("gpurun_out/prof_stats/*/*kernel_stats.csv")
ks = {}
if stats:
    for r in csv.DictReader(open(stats[0])):
        n = r["Name"]
        for k in ("xs_kernel", "spmv_kernel", "xs_finish_kernel", "finalize_kernel"):
            if k + "<" in n or k + "(" in n:
                ks[k] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6}
out["kernel_stats"] = ks
print(json.dumps(out, indent=1))
