#!/bin/bash
# r3: one emulated rank of P (private slots) under plan settings.  usage: tools/r3_mg_tune.sh <scale> <P> "ENV=VAL ..." ...
scale=$1; P=$2; shift 2
for e in "$@"; do
  [ "$e" = "-" ] && e="CGA_NONE=1"
  env $e CUGRAPH_AMD_MG_REGIONS=0 timeout -k 10 300 python3 -u tools/rank_emulate.py $scale $P 0 private 2>&1 | grep "^scale" | sed "s/^/[$e] /; s/ -> .*//"
done
