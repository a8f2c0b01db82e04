#!/bin/bash
# r3: set-up time of the multi-GPU driver for ONE rank's share at scale 28 over 8 ranks (537 M edges, 33.5 M rows):
# the driver at world size 1 with the collectives on, scale 25 (same edge and row counts).
for ex in gather personal; do
CUGRAPH_AMD_MG_EXCHANGE=$ex CUGRAPH_AMD_MG_PROFILE=1 timeout -k 10 300 python3 -u bench.py --scale ${1:-25} --force-mg --steps 5 --warmup 2 --no-cpu-baseline --no-convergence --no-extra-configs > gpurun_out/r3_mgb_$ex.json 2> gpurun_out/r3_mgb_$ex.err || { tail -20 gpurun_out/r3_mgb_$ex.err; exit 1; }
grep "MGPROF" gpurun_out/r3_mgb_$ex.err
python3 -c "
import json
d=json.load(open('gpurun_out/r3_mgb_$ex.json'))
print('$ex: t_build %.3f s  ms/step %.4f'%(d['graph_build']['coo_to_csc_s'], d['ms_per_step']))"
done
