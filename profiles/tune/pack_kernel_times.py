"""Device packer on the BASELINE batch: wall time of from_arcs_device and (under rocprofv3 --kernel-trace --stats) the kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from nfst_amd import synth
from nfst_amd.lattice import LatticeBatch
dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lats = synth.bench_batch(B)
n_rows, arc_off, src, label, dst, w = synth.batch_arcs(lats)
sd, ld, dd = (torch.from_numpy(x).to(dev) for x in (src, label, dst))
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lat = LatticeBatch.from_arcs_device(n_rows, arc_off, sd, ld, dd, 256, device=dev)
    torch.cuda.synchronize(); print("from_arcs_device ms", (time.perf_counter() - t0) * 1e3, flush=True)
lats = synth.snips_shaped_batch(64)
n_rows, arc_off, src, label, dst, w = synth.batch_arcs(lats)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lat = LatticeBatch.from_arcs_device(n_rows, arc_off, src, label, dst, 250, device=dev)
    torch.cuda.synchronize(); print("snips 64 from_arcs_device (with upload) ms", (time.perf_counter() - t0) * 1e3, flush=True)
