#!/bin/bash
# two ranks of the driver sharing GPU 0 over gloo, with a stack dump if a rank stops making progress
export CUGRAPH_AMD_TEST_SHARED_GPU=1 CUGRAPH_AMD_MG_REGIONS=${REGIONS:-0} CUGRAPH_AMD_XS_MIN_EDGES=65536 CUGRAPH_AMD_TEST_DUMP_AFTER=60
timeout -k 10 150 python -u -m torch.distributed.run --nnodes=1 --nproc-per-node=${WORLD:-2} --master-addr 127.0.0.1 --master-port 29611 tests/mg_world_worker.py 17 ${MODE:-unweighted} > gpurun_out/mg_shared.log 2>&1
echo "rc=$?" >> gpurun_out/mg_shared.log
tail -60 gpurun_out/mg_shared.log
