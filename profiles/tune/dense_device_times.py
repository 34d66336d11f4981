"""Dense tables on the device -> packed batch: where the time goes (32 BASELINE examples, collated)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from nfst_amd import synth
from nfst_amd.lattice import LatticeBatch
dev = torch.device("cuda")
lats = synth.bench_batch(32)
em, tr = synth.collate_dense([l.dense() for l in lats])
em_d, tr_d = torch.from_numpy(em).to(dev), torch.from_numpy(tr).to(dev)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lat = LatticeBatch.from_dense_device(em_d, tr_d)
    torch.cuda.synchronize(); print("from_dense_device ms", (time.perf_counter() - t0) * 1e3, "tables MB", (em.nbytes + tr.nbytes) / 1e6, flush=True)
