#!/bin/bash
# r3: {layout x P x scale} -> per-rank compute and bytes received, one rank emulated on one GPU (tools/rank_emulate.py)
set -o pipefail
out=gpurun_out/r3_mg_table.txt; : > $out
for sc in 26 28; do for P in 2 4 8; do
  CUGRAPH_AMD_MG_REGIONS=0 timeout -k 10 600 python3 -u tools/rank_emulate.py $sc $P 0 2>&1 | grep "^scale\|^rank" | sed "s/^/[shared] /" >> $out || exit 1
  CUGRAPH_AMD_MG_REGIONS=0 timeout -k 10 600 python3 -u tools/rank_emulate.py $sc $P 0 private 2>&1 | grep "^scale\|^rank" | sed "s/^/[private] /" >> $out || exit 1
  if [ $P -ge 4 ]; then
    CUGRAPH_AMD_MG_REGIONS=1 timeout -k 10 600 python3 -u tools/rank_emulate.py $sc $P 0 2>&1 | grep "^scale\|^rank" | sed "s/^/[two regions] /" >> $out || exit 1
  fi
done; done
cat $out
