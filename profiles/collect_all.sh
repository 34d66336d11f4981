#!/bin/bash
# Everything under profiles/ that is measured on the GPU box, in one call (round tag as argument): run through gpurun from the
# repo root, then copy gpurun_out/<tag>_* into profiles/.
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
S=profiles/tune/steps.sh
$S "timeout -k 10 500 bash profiles/collect.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1; tail -2 gpurun_out/${TAG}_collect.log | cut -c1-400" \
   "timeout -k 10 400 bash profiles/collect_sq.sh > gpurun_out/${TAG}_collect_sq.log 2>&1; tail -3 gpurun_out/${TAG}_collect_sq.log" \
   "NFST_TUNING=1 NFST_LIB=nfst_amd/lib/variants/libnfst_hip_prof.so timeout -k 10 200 python profiles/tune/stamps.py 256 4 > gpurun_out/${TAG}_stage_stamps.txt 2>&1; NFST_TUNING=1 NFST_LIB=nfst_amd/lib/variants/libnfst_hip_prof.so timeout -k 10 200 python profiles/tune/stamps.py 256 1 >> gpurun_out/${TAG}_stage_stamps.txt 2>&1; grep -c . gpurun_out/${TAG}_stage_stamps.txt" \
   "timeout -k 10 300 python profiles/bench_ingest.py 64 > gpurun_out/${TAG}_ingest.json 2> gpurun_out/${TAG}_ingest.err; tail -c 300 gpurun_out/${TAG}_ingest.json" \
   "timeout -k 10 400 python profiles/bench_secondary.py > gpurun_out/${TAG}_secondary_kernels.json 2> gpurun_out/${TAG}_secondary.err; tail -c 300 gpurun_out/${TAG}_secondary_kernels.json" \
   "timeout -k 10 400 bash profiles/collect_secondary.sh > gpurun_out/${TAG}_collect_secondary.log 2>&1; tail -3 gpurun_out/${TAG}_collect_secondary.log" \
   "timeout -k 10 300 python profiles/tune/deep_times.py > gpurun_out/${TAG}_deep_lattices.txt 2>&1; cat gpurun_out/${TAG}_deep_lattices.txt | grep -v amdgpu" \
   "timeout -k 10 300 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 200 gpurun_out/${TAG}_bench.json"
