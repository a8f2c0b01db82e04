"""Fold the passes of tools/profile_r3.sh (gpurun_out/r3_*) into profiles/r3_pmc_xs.json: per-launch averages of the
dominant kernel (xs2_kernel / xs_kernel) for the headline config and for C2 / C3, the FETCH_SIZE calibration and the
corrected HBM traffic.  usage: r3_pmc_fold.py <commit the passes were taken at>"""
import csv, glob, json, collections, re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
commit = sys.argv[1] if len(sys.argv) > 1 else subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()

def newest(pattern):
    """gpurun merges a call's files into gpurun_out/ beside those of earlier calls: take the latest"""
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)

def per_kernel(d, pat=r"xs2?_kernel<"):
    f = [newest(f"gpurun_out/{d}/**/*counter_collection.csv")]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    names = set()
    for r in csv.DictReader(open(f[0])):
        if re.search(pat, r["Kernel_Name"]):
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
            names.add(re.sub(r"\(.*", "", r["Kernel_Name"].replace("void cga::(anonymous namespace)::", "")))
    out = {c: sum(v.values()) / len(v) for c, v in acc.items()}
    return out, (len(next(iter(acc.values()))) if acc else 0), sorted(names)

def stats(d):
    ks = {}
    for r in csv.DictReader(open(newest(f"gpurun_out/{d}/**/*kernel_stats.csv"))):
        for k in ("xs2_kernel", "xs_kernel", "spmv_kernel", "xs_finish_kernel", "finalize_kernel", "pers_fix_kernel"):
            if "::" + k + "<" in r["Name"] or "::" + k + "(" in r["Name"]:
                ks[k] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6}
    return ks

KiB = 1024.0
fetch, n, kern = per_kernel("r3_fetch")
write, _, _ = per_kernel("r3_write")
tcc, _, _ = per_kernel("r3_tcc")
sq, _, _ = per_kernel("r3_sq")
cal, _, _ = per_kernel("r3_cal_xs")
b26 = json.load(open("gpurun_out/r3_stats.json"))
alg26 = b26["roofline"]["algorithmic_bytes_per_launch"]
# the entry stream of the headline config: E_xs * 4 (session info: algorithmic bytes minus row offsets and the vector)
stream26 = 1051449029 * 4
ratio = cal["FETCH_SIZE"] * KiB / stream26
def traffic(fetch_kib, write_kib, stream):
    read = fetch_kib * KiB + stream * (1.0 - ratio)
    return read, write_kib * KiB, read + write_kib * KiB
r26, w26, t26 = traffic(fetch["FETCH_SIZE"], write["WRITE_SIZE"], stream26)
ks26 = stats("r3_stats")
dom = "xs2_kernel" if "xs2_kernel" in ks26 else "xs_kernel"
out = {
    "commit": commit,
    "command": "rocprofv3 --pmc <one counter set per pass> --output-format csv -- python3 bench.py --scale 26 --steps 5 --warmup 2 "
               "--no-cpu-baseline --no-convergence --no-extra-configs  (tools/profile_r3.sh)",
    "scale": 26, "edge_factor": 16, "kernel": kern, "launches_averaged": n,
    "FETCH_SIZE_KiB": fetch["FETCH_SIZE"], "WRITE_SIZE_KiB": write["WRITE_SIZE"],
    "TCC_EA0_RDREQ": tcc.get("TCC_EA0_RDREQ_sum"), "TCC_EA0_RDREQ_32B": tcc.get("TCC_EA0_RDREQ_32B_sum"),
    "TCC_HIT": tcc.get("TCC_HIT_sum"), "TCC_MISS": tcc.get("TCC_MISS_sum"), "SQ": sq,
    "calibration": {
        "known_read_bytes": stream26, "FETCH_SIZE_KiB": cal["FETCH_SIZE"], "ratio": ratio,
        "note": "timing-aid build (make AIDS=1), CUGRAPH_AMD_BUCKET_MASK=32 CUGRAPH_AMD_GATHER_MASK=0xFF: only the 4 B/lane entry "
                "stream reaches the fabric; coalesced streams are tallied at about one half on gfx950 (MI355X_MICROARCH.md), the "
                "correction adds the part of the stream the calibration shows uncounted; gather misses are taken as counted"},
    "streamed_read_bytes": stream26, "read_bytes_corrected": r26, "write_bytes": w26,
    "traffic_bytes_per_launch": t26, "algorithmic_bytes_per_launch": alg26,
    "kernel_avg_ms_rocprof_kernel_trace": ks26[dom]["avg_ms"], "kernel_stats": ks26, "other_configs": {},
}
# C2 / C3: the same correction with the stream taken as the share of the algorithmic bytes it has at scale 26
share = stream26 / alg26
c2 = json.load(open("gpurun_out/r3_c2_stats.json"))
c2f, c2n, c2k = per_kernel("r3_c2_fetch")
c2w, _, _ = per_kernel("r3_c2_write")
alg = c2["roofline"]["algorithmic_bytes_per_launch"]
r, w, t = traffic(c2f["FETCH_SIZE"], c2w["WRITE_SIZE"], alg * share)
out["other_configs"]["C2_scale22_fp32_unweighted"] = {
    "command": "... -- python3 bench.py --scale 22 --scramble 0 --steps 5 --warmup 2 --no-cpu-baseline --no-convergence --no-extra-configs",
    "kernel": c2k, "launches_averaged": c2n, "FETCH_SIZE_KiB": c2f["FETCH_SIZE"], "WRITE_SIZE_KiB": c2w["WRITE_SIZE"],
    "algorithmic_bytes_per_launch": alg, "streamed_read_bytes_estimate": alg * share, "read_bytes_corrected": r,
    "write_bytes": w, "traffic_bytes_per_launch": t, "kernel_stats": stats("r3_c2_stats")}
c3line = [l for l in open("gpurun_out/r3_c3_stats.txt") if l.startswith("C3JSON ")]
if c3line:
    c3 = json.loads(c3line[-1][7:])
    c3f, c3n, c3k = per_kernel("r3_c3_fetch")
    c3w, _, _ = per_kernel("r3_c3_write")
    alg = c3["dominant_kernel_bytes"]
    r, w, t = traffic(c3f["FETCH_SIZE"], c3w["WRITE_SIZE"], alg * share)
    out["other_configs"]["C3_scale24_fp64_weighted_personalized_1k"] = {
        "command": "... -- python3 tools/c3_bench.py 24", "kernel": c3k, "launches_averaged": c3n,
        "FETCH_SIZE_KiB": c3f["FETCH_SIZE"], "WRITE_SIZE_KiB": c3w["WRITE_SIZE"], "algorithmic_bytes_per_launch": alg,
        "streamed_read_bytes_estimate": alg * share, "read_bytes_corrected": r, "write_bytes": w,
        "traffic_bytes_per_launch": t, "kernel_stats": stats("r3_c3_stats")}
json.dump(out, open("profiles/r3_pmc_xs.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k not in ("SQ", "kernel_stats")}, indent=1))
