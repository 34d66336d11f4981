#!/bin/bash
# product vs a variant library at several batch sizes, one box
R=${GRAFT_REPO_ROOT:-$(pwd)}
for B in ${SIZES:-256 512 1024 2048}; do
  for v in "" ${VARIANTS:-base}; do
    lib=""; [ -n "$v" ] && lib=$R/nfst_amd/lib/variants/libnfst_hip_$v.so
    for D in ${DIRECTS:-0}; do
    echo -n "B=$B variant=${v:-product} direct=$D "
    NFST_TUNE_DIRECT=$D NFST_LIB=$lib python bench.py --no-aux --no-cpu-baseline --steps 200 --lattices-per-gpu $B 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('   ms', round(d['ms_per_step'],5), 'kern', round(d['roofline']['kernel_ms'],5), 'frac', round(d['roofline']['frac'],4), 'Garcs/s', round(d['value']/1e9,1))"
    done
  done
done
