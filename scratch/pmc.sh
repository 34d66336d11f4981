#!/bin/bash
# PMC counters of the backward-only kernel (instruction mix / waits per dispatch)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
run() { rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$1 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --mode bwd > /dev/null 2>$OUT/err_$1.log; }
run SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_WAIT_ANY
run SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/SQ_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_backward" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(k, "per dispatch:", sum(v)/len(v), "n", len(v))
PY
