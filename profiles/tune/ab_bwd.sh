#!/bin/bash
# beta-only sweep: tile waves (default) vs loader + decoder + sweep (NFST_TW=0)
run() { echo -n "env='$1' args='$2' "
  env $1 timeout -k 5 120 python bench.py --no-aux --no-cpu-baseline --steps 400 $2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('   ms', round(d['ms_per_step'],5), 'kern', round(d['roofline']['kernel_ms'],5))"; }
for a in "--mode bwd" "--mode bwd --arc-scores" "--mode bwd --width 4"; do
  run "NFST_TW=0" "$a"; run "NFST_TW=1" "$a"
done
