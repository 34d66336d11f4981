#!/bin/bash
# Round 3: the chunked flavour for deep, narrow lattices -- timings against the general kernels,
# per-phase stamps of the sweep kernel, rocprofv3 kernel stats of 200 launches on the SNIPS-shaped batch of 64.
# Run on the GPU box from the repo root: bash profiles/collect_chunked.sh  (writes gpurun_out/chunked/)
set -e
OUT=$PWD/gpurun_out/chunked
mkdir -p $OUT
python -m nfst_amd.build --variant prof -DNFST_PROF > $OUT/build_prof.log 2>&1
python profiles/tune/chunk_times.py > $OUT/times.txt 2> $OUT/times.err
NFST_TUNING=1 NFST_LIB=nfst_amd/lib/variants/libnfst_hip_prof.so python profiles/tune/chunk_stamps.py 64 > $OUT/stamps_b64.txt 2> $OUT/stamps.err
NFST_TUNING=1 NFST_LIB=nfst_amd/lib/variants/libnfst_hip_prof.so python profiles/tune/chunk_stamps.py 16 > $OUT/stamps_b16.txt 2>> $OUT/stamps.err
NFST_TUNING=1 NFST_LIB=nfst_amd/lib/variants/libnfst_hip_prof.so python profiles/tune/chunk_stamps.py 64 narrow > $OUT/stamps_narrow_b64.txt 2>> $OUT/stamps.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o chk -- python3 $GRAFT_REPO_ROOT/profiles/tune/chunk_prof.py > $OUT/prof.log 2>&1
cp $OUT/prof/chk_kernel_stats.csv $OUT/kernel_stats.csv
rm -rf $OUT/prof
