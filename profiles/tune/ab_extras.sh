#!/bin/bash
# per-arc scores inside the kernel (extras waves): placement / priority variants against the plain step, one box
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { # variant, bench args
  lib=""; [ -n "$1" ] && lib=$R/nfst_amd/lib/variants/libnfst_hip_$1.so
  echo -n "variant=${1:-product} args='$2' "
  NFST_LIB=$lib timeout -k 5 120 python bench.py --no-aux --no-cpu-baseline --steps 300 $2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('   ms', round(d['ms_per_step'],5), 'kern', round(d['roofline']['kernel_ms'],5))"
}
run "" ""
run "" "--mode fb_sweeps_only"
for v in ${VARIANTS:-"" xw1 xw2}; do
  run "$v" "--arc-scores"
  run "$v" "--arc-scores --mode fb_sweeps_only"
done
run "" "--arc-scores --mode bwd"
run "" "--mode bwd"
run "" "--arc-scores --lattices-per-gpu 1024"
run "" "--lattices-per-gpu 1024"
