R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_sec_r03
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/profiles/bench_secondary.py > $OUT/out.json 2> $OUT/err.txt
python3 - <<PY
import csv, glob
st = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(st[0])))
open("$R/gpurun_out/r03_secondary_kernel_stats.csv", "w").write(open(st[0]).read())
for r in rows[:30]:
    print(r["Name"][:110], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
