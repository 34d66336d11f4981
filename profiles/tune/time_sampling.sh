#!/bin/bash
# posterior sampling (k_sample with the lattice's CSR staged in LDS) on the BASELINE batch and on configs[4]-shaped lattices
python - <<'PY'
import time, torch, numpy as np
from nfst_amd import ops, synth
from nfst_amd.lattice import LatticeBatch
dev = torch.device("cuda:0")
def t(f, n=100):
    for _ in range(10): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
lat = LatticeBatch.from_synth(synth.bench_batch(256), device=dev); th = torch.from_numpy(synth.label_scores(1, 256)).to(dev)
beta = ops.backward(lat, th, want_logbeta=False, want_me=True)
for K in (1, 16, 64, 128):
    print("baseline batch K", K, "k_sample ms", round(t(lambda: ops.sample_paths(lat, th, K, seed=1, beta=beta)), 4), " with the beta sweep", round(t(lambda: ops.sample_paths(lat, th, K, seed=1)), 4))
rng = np.random.default_rng(4); V = 64
lats = [synth.edit_lattice(rng.integers(3, V, size=int(rng.integers(6, 15))).tolist(), rng.integers(3, V, size=int(rng.integers(6, 15))).tolist(), vocab=V, seed=100 + i) for i in range(256)]
lat2 = LatticeBatch.from_synth(lats, device=dev); th2 = torch.from_numpy(synth.label_scores(12, V, mean=-1.0, std=0.7)).to(dev)
print("edit lattices K 16 ms", round(t(lambda: ops.sample_paths(lat2, th2, 16, seed=1)), 4))
PY
