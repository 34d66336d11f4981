#!/bin/bash
# burst / raw-slot sweep of the deep flavour (B=256) 
for cfg in "12 1" "16 2" "16 4" "20 4" "24 4" "24 8" "32 8"; do
  set -- $cfg
  echo "RS=$1 BURST=$2" 
  NFST_TUNE_RS=$1 NFST_TUNE_BURST=$2 python bench.py --no-aux --no-cpu-baseline --steps 400 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('  B256 ms', round(d['ms_per_step'],5), 'kern', round(d['roofline']['kernel_ms'],5), 'frac', round(d['roofline']['frac'],4))"
done
