#!/bin/bash
# usage: scratch/sweep.sh  -> prints kernel ms for a grid of settings
run() { timeout -k 10 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', '| depth', d['config']['max_depth'], '| kernel_ms %.4f' % d['roofline']['kernel_ms'], '| frac %.3f' % d['roofline']['frac'], '| ms/step %.4f' % d['ms_per_step'])"; }
for W in 1 2 4; do for m in fb fb_sweeps_only bwd; do run --sweep-waves $W --mode $m; done; done
for W in 1 2 4; do run --sweep-waves $W --width 4 --mode fb; run --sweep-waves $W --width 64 --mode fb; done
for W in 1 4; do run --sweep-waves $W --lattices-per-gpu 1024 --mode fb; done
