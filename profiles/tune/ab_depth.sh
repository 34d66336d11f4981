#!/bin/bash
# DMA depth variants (tiles in flight / staging slots of the deep flavour), B=256, one box
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for v in "" a16r20 a16r24 a24r28 a24r32 a32r36; do
  lib=""; [ -n "$v" ] && lib=$R/nfst_amd/lib/variants/libnfst_hip_$v.so
  for mode in fb fb_sweeps_only; do
  echo -n "variant=${v:-product} mode=$mode "
  NFST_LIB=$lib python bench.py --no-aux --no-cpu-baseline --steps 400 --mode $mode 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('   ms', round(d['ms_per_step'],5), 'kern', round(d['roofline']['kernel_ms'],5), 'maxdepth/tiles', d['config']['max_depth'], d['config']['max_tiles'])"
  done
done
done
