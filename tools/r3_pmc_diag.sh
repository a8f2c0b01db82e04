#!/bin/bash
# r3: diagnostic PMC passes on xs_kernel (one counter set per run, program directly after `--`):
# address translation (UTCL1), TA / TD / TCP stalls, instruction mix.  usage: tools/r3_pmc_diag.sh <tag> [scale]
tag=$1; scale=${2:-26}
B="bench.py --scale $scale --steps 5 --warmup 2 --no-cpu-baseline --no-convergence --no-extra-configs"
export TMPDIR=/tmp
pass() { # name, counters...
  local name=$1; shift
  rm -rf gpurun_out/${tag}_$name
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/${tag}_$name -- python3 $B > /dev/null 2>gpurun_out/${tag}_$name.err || { echo "$name-fail"; tail -3 gpurun_out/${tag}_$name.err; return 1; }
}
pass utcl TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE &&
pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum &&
pass utcl2 TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_TCC_READ_REQ_sum TA_TOTAL_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum &&
pass sqmix SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA &&
pass sqstall SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY &&
pass tcp2 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TAGRAM0_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum
python3 - "$tag" <<'PY'
import csv, glob, json, collections, sys
tag = sys.argv[1]
out = {}
for d in ("utcl", "tcp", "utcl2", "sqmix", "sqstall", "tcp2"):
    f = glob.glob(f"gpurun_out/{tag}_{d}/**/*counter_collection.csv", recursive=True)
    if not f:
        continue
    for match in ("xs_kernel", "spmv_kernel", "xs_finish_kernel"):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f[0])):
            if match + "<" in r["Kernel_Name"] or match + "(" in r["Kernel_Name"]:
                acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for c, v in acc.items():
            out.setdefault(match, {})[c] = sum(v.values()) / len(v)
json.dump(out, open(f"gpurun_out/{tag}_pmc_diag.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
echo diag-ok
