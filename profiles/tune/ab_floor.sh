#!/bin/bash
# where the B=256 step's time goes: ablation builds (nfst_amd/lib/variants) on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { # variant, bench args
  lib=""; [ -n "$1" ] && lib=$R/nfst_amd/lib/variants/libnfst_hip_$1.so
  echo -n "variant=${1:-product} args='$2' "
  NFST_LIB=$lib python bench.py --no-aux --no-cpu-baseline --steps 400 $2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('   ms', round(d['ms_per_step'],5), 'kern', round(d['roofline']['kernel_ms'],5))"
}
run "" ""
run "" "--mode fb_sweeps_only"
run nomath ""
run nomath "--mode fb_sweeps_only"
run decpass "--mode fb_sweeps_only"
run nosweep ""
run nosweep "--mode fb_sweeps_only"
run "" "--width 64"
run "" "--width 64 --mode fb_sweeps_only"
run "" "--mode bwd"
