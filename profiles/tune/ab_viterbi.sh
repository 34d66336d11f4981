#!/bin/bash
# Viterbi: tile-wave kernel (default) vs one wave reading the program from global memory (NFST_TW=0)
for tw in 0 1; do
NFST_TW=$tw python - <<'PY'
import os, time, torch, numpy as np
from nfst_amd import ops, synth
from nfst_amd.lattice import LatticeBatch
dev = torch.device("cuda:0")
def t(f, n=200):
    for _ in range(10): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
lat = LatticeBatch.from_synth(synth.bench_batch(256), device=dev); th = torch.from_numpy(synth.label_scores(1, 256)).to(dev)
asc = torch.randn(lat.total_arcs, device=dev) * 0.1
print("TW", os.environ["NFST_TW"], "baseline batch ms", round(t(lambda: ops.viterbi(lat, th)), 4), "with scores", round(t(lambda: ops.viterbi(lat, th, arc_scores=asc)), 4))
lats = synth.snips_shaped_batch(64); lat2 = LatticeBatch.from_synth(lats, device=dev); th2 = torch.from_numpy(synth.label_scores(64, lats[0].vocab, mean=-1.5, std=0.8)).to(dev)
print("   snips-shaped b64 ms", round(t(lambda: ops.viterbi(lat2, th2)), 4))
PY
done
