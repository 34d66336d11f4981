#!/bin/bash
for lib in nfst_amd/lib/libnfst_hip.so scratch/lib_NOWAIT.so scratch/lib_NOWAITDNFST_EXP_NODMA.so; do
  cp nfst_amd/lib/libnfst_hip.so /tmp/orig.so 2>/dev/null
  NFST_LIB=$lib python - <<'PY'
import os, ctypes as C, sys, json, subprocess
PY
  echo "== $lib"
  NFST_LIB_OVERRIDE=$lib timeout -k 10 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --mode bwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kernel_ms %.4f' % d['roofline']['kernel_ms'])"
done
