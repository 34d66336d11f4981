#!/bin/bash
run() { timeout -k 10 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', '| depth', d['config']['max_depth'], 'tiles', d['config']['max_tiles'], '| kernel_ms %.4f' % d['roofline']['kernel_ms'], '| frac %.3f' % d['roofline']['frac'], '| ms/step %.4f' % d['ms_per_step'])"; }
for m in fb fb_sweeps_only bwd; do run --mode $m; done
for U in 1 2 4; do run --slots $U --mode fb_sweeps_only; done
run --width 4 --mode fb; run --width 64 --mode fb
run --lattices-per-gpu 512 --mode fb
run --lattices-per-gpu 1024 --mode fb
run --lattices-per-gpu 2048 --mode fb
