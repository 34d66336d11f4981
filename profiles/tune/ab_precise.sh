#!/bin/bash
# A/B of the precise sweep loop's variants (variant libraries built with -DNFST_P_*), deep lattices
cd ${GRAFT_REPO_ROOT:-.}
export NFST_TUNING=1
for v in "" pa pb pc; do
  if [ -z "$v" ]; then unset NFST_LIB; echo "== product"; else export NFST_LIB=nfst_amd/lib/variants/libnfst_hip_$v.so; echo "== $v"; fi
  timeout -k 5 120 python profiles/tune/deep_times.py 2>&1 | grep "configs2\|width 4"
done
