#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
python -c "import torch"
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_sq
rm -rf $OUT && mkdir -p $OUT
CMD="python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-aux --no-replay"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/a -- $CMD > /dev/null 2> $OUT/a.err || { tail -5 $OUT/a.err; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/b -- $CMD > /dev/null 2> $OUT/b.err || { tail -5 $OUT/b.err; exit 1; }
python3 - <<PY
import csv, glob, json, collections
out = {}
for name in ("a", "b"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_forward_backward" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out[k] = {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)}
out["note"] = ("rocprofv3 --pmc, two passes of 8 SQ counters over bench.py --steps 50 --warmup 10, per k_forward_backward<1024,0,false,true> (tile waves) "
               "dispatch (256 workgroups x 16 waves); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves")
json.dump(out, open("$R/gpurun_out/r03_pmc_sq.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
