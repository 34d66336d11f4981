#!/bin/bash
# How the files in profiles/ are produced (run on the GPU box through gpurun from the repo root):
#   profiles/collect.sh r03
# 1. kernel trace + stats of the benchmark command  -> <tag>_kernel_stats.csv
# 2. HBM traffic counters, one pass each            -> <tag>_pmc_traffic.json
#    (MI355X_MICROARCH.md, "HBM": FETCH_SIZE and WRITE_SIZE are reported in KiB;
#     on gfx950 FETCH_SIZE reads half of a wide coalesced streaming read)
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
CMD="python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-aux --no-replay ${COLLECT_ARGS}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $CMD > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $CMD > /dev/null 2> $OUT/write.err
python3 - <<PY
import csv, glob, json, collections
out = {}
st = glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True)
if st:
    rows = list(csv.DictReader(open(st[0])))
    open("$R/gpurun_out/${TAG}_kernel_stats.csv", "w").write(open(st[0]).read())
    for r in rows[:6]:
        print(r)
for name in ("fetch", "write"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_forward_backward" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out[k] = {"mean_per_dispatch_KiB": sum(v) / len(v), "dispatches": len(v)}
out["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py --steps 50 --warmup 10; "
               "values in KiB per k_forward_backward dispatch; gfx950 FETCH_SIZE counts 64 B per 128-B streaming request")
# the workload the counters were taken on (bench.py attaches a traffic file only to the same workload)
try:
    line = [l for l in open("$OUT/bench_under_rocprof.json") if l.startswith("{")][-1]
    cfg = json.loads(line)["config"]
    out["lattices_per_gpu"] = cfg["lattices_per_gpu"]
    out["rotate"] = cfg.get("resident_batches_rotated", 1)
    out["cache_state"] = cfg.get("cache_state")
except Exception as e:
    out["workload_error"] = str(e)
json.dump(out, open("$R/gpurun_out/${TAG}_pmc_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
