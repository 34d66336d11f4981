"""Per-workgroup time stamps of k_forward_backward (instrumented build: python -m nfst_amd.build --variant prof -DNFST_PROF;
run with NFST_LIB=.../libnfst_hip_prof.so).  Stamps (100 MHz): 0 entry, 1 init done, 2 beta sweep done, 3 alpha sweep done,
4 after the barrier, 5 beta decoder done, 6 beta loader done, 7 end."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from nfst_amd import ops, synth, _lib
from nfst_amd.lattice import LatticeBatch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mode = sys.argv[2] if len(sys.argv) > 2 else "fb"
dev = torch.device("cuda")
lat = LatticeBatch.from_synth(synth.bench_batch(B), device=dev)
theta = torch.from_numpy(synth.label_scores(1, 256)).to(dev)
out = None
kw = dict(want_alpha_beta=mode == "fb", want_posterior=mode == "fb")
for _ in range(20):
    out = ops.forward_backward(lat, theta, out=out, **kw)
torch.cuda.synchronize()
raw = C.CDLL(_lib.LIB_PATH)
buf = np.zeros(B * 8, np.uint64)
assert raw.nfst_prof_read(buf.ctypes.data_as(C.c_void_p), B * 8) == 0
t = buf.reshape(B, 8).astype(np.int64)
t0 = t[:, 0].min()
rel = (t - t0) / 100.0  # us since the first workgroup's entry
tiles = lat.meta_host[:, _lib.META_BWD_TILES]
names = ["entry", "init", "beta_done", "alpha_done", "barrier", "bdec_done", "bload_done", "end"]
print(f"B={B} mode={mode}: us since the first workgroup entered (min / median / max over workgroups)")
for i, n in enumerate(names):
    print(f"  {n:10s} {rel[:, i].min():8.2f} {np.median(rel[:, i]):8.2f} {rel[:, i].max():8.2f}")
dur = (t[:, 2] - t[:, 1]) / 100.0
print("  beta sweep duration / tile (ns): min %.0f median %.0f max %.0f" % tuple(np.percentile(dur / tiles * 1e3, [0, 50, 100])))
ld = (t[:, 6] - t[:, 1]) / 100.0
print("  beta loader duration / tile (ns): min %.0f median %.0f max %.0f" % tuple(np.percentile(ld / tiles * 1e3, [0, 50, 100])))
print("  tiles: min %d median %d max %d" % (tiles.min(), np.median(tiles), tiles.max()))
slow = np.argsort(rel[:, 7])[-5:]
print("  slowest workgroups:", [(int(b), int(tiles[b]), round(float(rel[b, 1]), 2), round(float(rel[b, 2]), 2), round(float(rel[b, 7]), 2)) for b in slow])
