#!/bin/bash
# A/B of variant libraries (python -m nfst_amd.build --variant TAG -D...) against the product on the headline: cold and replayed
# kernel time.  usage: profiles/tune/ab_variants.sh tag1 tag2 ...   (BENCH_ARGS for extra bench.py flags)
cd ${GRAFT_REPO_ROOT:-.}
export NFST_TUNING=1
for v in "" "$@"; do
  if [ -z "$v" ]; then unset NFST_LIB; name=product; else export NFST_LIB=nfst_amd/lib/variants/libnfst_hip_$v.so; name=$v; fi
  for rep in 1 2; do
  timeout -k 5 200 python bench.py --no-aux --no-cpu-baseline --steps 400 ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$name', 'cold %.2f us  replay %.2f us  frac %.3f' % (r['kernel_ms_cold']*1e3, r['kernel_ms_replay']*1e3, r['frac']))"
  done
done
