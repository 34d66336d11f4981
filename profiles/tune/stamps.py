"""Per-workgroup time stamps of the tile-wave k_forward_backward (instrumented build: python -m nfst_amd.build --variant prof
-DNFST_PROF; run with NFST_TUNING=1 NFST_LIB=.../libnfst_hip_prof.so).  Stamps (100 MHz): 0 entry, 1 init done, 2 beta sweep done,
3 alpha sweep done, 4 after the barrier, 5 the beta sweep has its first decoded tile, 6 first beta tile wave done, 7 end.
python profiles/tune/stamps.py [lattices] [rotate]: `rotate` distinct resident batches take turns (cold launches, as bench.py)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from nfst_amd import ops, synth, _lib
from nfst_amd.lattice import LatticeBatch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rot = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda")
lats = [LatticeBatch.from_synth(synth.bench_batch(B, first_seed=1234 + 100000 * r), device=dev) for r in range(rot)]
theta = torch.from_numpy(synth.label_scores(1, 256)).to(dev)
outs = [None] * rot
for it in range(10 * rot + 1):  # the last launch is batch 0 again, after the others have passed through the caches
    r = it % rot
    outs[r] = ops.forward_backward(lats[r], theta, out=outs[r])
torch.cuda.synchronize()
raw = C.CDLL(_lib.LIB_PATH)
buf = np.zeros(B * 8, np.uint64)
assert raw.nfst_prof_read(buf.ctypes.data_as(C.c_void_p), B * 8) == 0
t = buf.reshape(B, 8).astype(np.int64)
rel = (t - t[:, 0].min()) / 100.0  # us since the first workgroup's entry
lat = lats[0]
tiles = np.maximum(lat.meta_host[:, _lib.META_BWD_TILES], lat.meta_host[:, _lib.META_FWD_TILES])
names = ["entry", "init done", "beta swept", "alpha swept", "barrier", "first tile", "tile wave 0 done", "end"]
print(f"B={B}, {rot} batches rotating: us since the first workgroup entered (min / median / max over workgroups)")
for i in (0, 1, 5, 6, 2, 3, 4, 7):
    print(f"  {names[i]:18s} {rel[:, i].min():8.2f} {np.median(rel[:, i]):8.2f} {rel[:, i].max():8.2f}")
own = (t - t[:, :1]) / 100.0
print("  per workgroup, since its own entry: init %.2f, first tile %.2f, sweeps done %.2f, barrier -> end %.2f (medians)" %
      (np.median(own[:, 1]), np.median(own[:, 5]), np.median(np.maximum(own[:, 2], own[:, 3])), np.median(own[:, 7] - own[:, 4])))
dur = (np.maximum(t[:, 2], t[:, 3]) - t[:, 5]) / 100.0
print("  sweep duration / tile (ns): min %.0f median %.0f max %.0f;  tiles: min %d median %d max %d" %
      (*np.percentile(dur / tiles * 1e3, [0, 50, 100]), tiles.min(), np.median(tiles), tiles.max()))
slow = np.argsort(rel[:, 7])[-4:]
print("  slowest workgroups (id, tiles, first tile, swept, end):", [(int(b), int(tiles[b]), round(float(rel[b, 5]), 2), round(float(max(rel[b, 2], rel[b, 3])), 2), round(float(rel[b, 7]), 2)) for b in slow])
