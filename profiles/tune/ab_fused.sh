#!/bin/bash
# fused sweeps (default for all-compact batches) vs the loader / decoder / sweep pipeline (NFST_NO_FUSED=1), one box
for B in ${SIZES:-256 512 1024 2048}; do
  for nf in "" 1; do
    echo -n "B=$B pipeline=$([ -n "$nf" ] && echo three-wave || echo fused) "
    NFST_NO_FUSED=${nf:-0} timeout -k 10 120 python bench.py --no-aux --no-cpu-baseline --steps 200 --lattices-per-gpu $B ${BENCH_ARGS} 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('   ms', round(d['ms_per_step'],5), 'kern', round(d['roofline']['kernel_ms'],5), 'frac', round(d['roofline']['frac'],4), 'Garcs/s', round(d['value']/1e9,1))"
  done
done
