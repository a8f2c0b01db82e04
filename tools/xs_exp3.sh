#!/bin/bash
B="bench.py --scale 26 --steps 3 --warmup 1 --no-cpu-baseline --no-convergence"
export TMPDIR=/tmp
export CUGRAPH_AMD_BUCKET_MASK=32
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b
timeout -s KILL 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_a -- python3 $B > /dev/null 2>&1 || echo a-fail
timeout -s KILL 300 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_b -- python3 $B > /dev/null 2>&1 || echo b-fail
echo pmc-ok
