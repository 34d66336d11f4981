#!/bin/bash
# tile-wave placement / priority variants (libraries built with -DNFST_TW_PLACE=0, -DNFST_TW_PRIO)
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { lib=""; [ -n "$1" ] && [ "$1" != product ] && lib=$R/nfst_amd/lib/variants/libnfst_hip_$1.so
  echo -n "variant=${1:-product} env='$3' args='$2' "
  env $3 NFST_LIB=$lib timeout -k 5 120 python bench.py --no-aux --no-cpu-baseline --steps 400 $2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('   ms', round(d['ms_per_step'],5), 'kern', round(d['roofline']['kernel_ms'],5))"; }
for a in "" "--mode fb_sweeps_only" "--width 4" "--width 64" "--arc-scores"; do
  run "" "$a" "NFST_TW=0"
  for v in ${VARIANTS:-product}; do run "$v" "$a" "NFST_TW=1"; done
done
