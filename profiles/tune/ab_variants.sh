#!/bin/bash
# B=256 step under several builds of the library (nfst_amd/lib/variants), same box, interleaved twice
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for v in "" base nomath nogather $EXTRA_VARIANTS; do
  lib=""; [ -n "$v" ] && lib=$R/nfst_amd/lib/variants/libnfst_hip_$v.so
  echo -n "variant=${v:-product} "
  NFST_LIB=$lib python bench.py --no-aux --no-cpu-baseline --steps 400 ${BENCH_ARGS} 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('   ms', round(d['ms_per_step'],5), 'kern', round(d['roofline']['kernel_ms'],5), 'frac', round(d['roofline']['frac'],4))"
done
done
