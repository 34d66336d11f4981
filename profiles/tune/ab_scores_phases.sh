run() { echo -n "args='$1' "
  timeout -k 5 120 python bench.py --no-aux --no-cpu-baseline --steps 300 $1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('   ms', round(d['ms_per_step'],5), 'kern', round(d['roofline']['kernel_ms'],5))"; }
run ""
run "--mode fb_sweeps_only"
run "--arc-scores"
run "--arc-scores --mode fb_sweeps_only"
run "--arc-scores --mode bwd"
run "--mode bwd"
