#!/bin/bash
# true kernel durations (rocprofv3 kernel trace) of the ablation builds
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "" nomath decpass nosweep; do
  lib=""; [ -n "$v" ] && lib=$R/nfst_amd/lib/variants/libnfst_hip_$v.so
  for mode in fb fb_sweeps_only; do
    out=$R/gpurun_out/prof_var/${v:-product}_$mode
    rm -rf $out; mkdir -p $out
    NFST_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $R/bench.py --no-aux --no-cpu-baseline --steps 100 --warmup 10 --mode $mode > /dev/null 2> $out/err.txt
    f=$(find $out -name "*kernel_stats.csv" | head -1)
    echo "variant=${v:-product} mode=$mode: $(grep k_forward_backward $f | head -1 | cut -d, -f1-8 | sed 's/_ZN12_GLOBAL__N_1//')"
  done
done
