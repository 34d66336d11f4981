#!/bin/bash
# A/B of the direct flavour (sweep waves decode for themselves; NFST_TUNE_DIRECT=1) at several batch sizes
for B in 512 1024 2048; do
  for D in 0 1; do
    echo "B=$B DIRECT=$D"
    NFST_TUNE_DIRECT=$D python bench.py --no-aux --no-cpu-baseline --steps 200 --lattices-per-gpu $B | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('   ms', round(d['ms_per_step'],5), 'kern', round(d['roofline']['kernel_ms'],5), 'frac', round(d['roofline']['frac'],4), 'Garcs/s', round(d['value']/1e9,1))"
  done
done
