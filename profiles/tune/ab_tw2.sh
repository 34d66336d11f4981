# two lattices per CU: tile-wave kernel (two workgroups per CU, rings of four slots) against the fused sweeps
R=${GRAFT_REPO_ROOT:-$(pwd)}
for n in 320 384 512; do
  for v in 0 1; do
    echo "lattices=$n NFST_TW2=$v"
    NFST_TW2=$v timeout -k 5 120 python3 $R/bench.py --lattices-per-gpu $n --no-aux --no-cpu-baseline --steps 100 | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value']/1e9,1), 'G arcs/s', round(d['roofline']['kernel_ms']*1e3,1), 'us', round(d['roofline']['frac'],3))" || exit 1
  done
done
