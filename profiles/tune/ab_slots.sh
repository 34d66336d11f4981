run() { echo -n "env='$1' args='$2' "
  env $1 timeout -k 5 120 python bench.py --no-aux --no-cpu-baseline --steps 300 $2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('   ms', round(d['ms_per_step'],5), 'kern', round(d['roofline']['kernel_ms'],5), 'tiles', d['config']['max_tiles'], 'depth', d['config']['max_depth'])"; }
for w in 4 8 16 64; do
run "A=1" "--width $w"
run "A=1" "--width $w --slots 4"
done
