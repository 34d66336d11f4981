#!/bin/bash
# weight waves for the plain step (NFST_WW=1: idle waves gather the label weights, the decoder only writes addresses)
run() { echo -n "env='$1' args='$2' "
  env $1 timeout -k 5 120 python bench.py --no-aux --no-cpu-baseline --steps 400 $2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('   ms', round(d['ms_per_step'],5), 'kern', round(d['roofline']['kernel_ms'],5), 'loss', d['config']['loss'])"; }
for rep in 1 2; do
for a in "" "--mode fb_sweeps_only" "--mode bwd" "--width 4" "--width 64"; do
  run "NFST_WW=0" "$a"
  run "NFST_WW=1" "$a"
done
done
