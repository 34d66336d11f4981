#!/bin/bash
# headline shapes under the tuning knobs given in the environment
for B in 256 512 1024 2048; do
  for D in 0 1; do
    if [ $B = 256 ] && [ $D = 1 ]; then continue; fi
    echo -n "B=$B DIRECT=$D "
    NFST_TUNE_DIRECT=$D python bench.py --no-aux --no-cpu-baseline --steps 200 --lattices-per-gpu $B 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('   ms', round(d['ms_per_step'],5), 'kern', round(d['roofline']['kernel_ms'],5), 'frac', round(d['roofline']['frac'],4), 'Garcs/s', round(d['value']/1e9,1))"
  done
done
