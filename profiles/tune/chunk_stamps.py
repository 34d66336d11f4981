"""Per-workgroup stamps of k_chunk_sweep (instrumented build: python -m nfst_amd.build --variant prof -DNFST_PROF; run with
NFST_TUNING=1 NFST_LIB=nfst_amd/lib/variants/libnfst_hip_prof.so).  Stamps (100 MHz): 0 entry, 1 weights + init done,
2 pass 1 done, 3 pass 2 done, 4 end.  python profiles/tune/chunk_stamps.py [lattices]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from nfst_amd import ops, synth, _lib
from nfst_amd.lattice import LatticeBatch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda")
if len(sys.argv) > 2 and sys.argv[2] == "narrow":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(8000)
    lats = [synth.layered_lattice(8000 + i, n_states=int(rng.integers(400, 1501)), avg_degree=float(rng.choice([3.0, 5.0])), vocab=250,
                                  width=int(rng.choice([1, 2, 3])), span=int(rng.choice([1, 2])), max_degree=40) for i in range(B)]
else:
    lats = synth.snips_shaped_batch(B, vocab=250)
host = LatticeBatch.from_synth(lats)
assert host.build_chunks(force=True)
lat = host.to(dev)
theta = torch.from_numpy(synth.label_scores(1, 250, mean=-1.5, std=0.8)).to(dev)
for _ in range(5):
    r = ops.forward_backward(lat, theta)
torch.cuda.synchronize()
raw = C.CDLL(_lib.LIB_PATH)
n = 2 * B
buf = np.zeros(n * 8, np.uint64)
assert raw.nfst_prof_read(buf.ctypes.data_as(C.c_void_p), n * 8) == 0
t = buf.reshape(n, 8).astype(np.int64)
own = (t[:, 1:5] - t[:, :4]) / 100.0
rel_end = (t[:, 4] - t[:, 0].min()) / 100.0
m = lat.chunks.meta_host.reshape(n, 8)
tab = lat.chunks._t["tab"].cpu().numpy().reshape(-1, 4)
longest = np.array([tab[m[i, 4]:m[i, 4] + m[i, 0], 2].max() for i in range(n)])
print("us per phase (median / max over workgroups): weights+init %.1f / %.1f, pass 1 %.1f / %.1f, pass 2 %.1f / %.1f, pass 3 %.1f / %.1f; last end %.1f" %
      (np.median(own[:, 0]), own[:, 0].max(), np.median(own[:, 1]), own[:, 1].max(), np.median(own[:, 2]), own[:, 2].max(),
       np.median(own[:, 3]), own[:, 3].max(), rel_end.max()))
print("pass 1: ns per entry of the longest chunk: median %.0f min %.0f max %.0f" % tuple(np.percentile(own[:, 1] * 1e3 / longest, [50, 0, 100])))
print("pass 2: ns per chunk: median %.0f; by F:" % np.median(own[:, 2] * 1e3 / m[:, 0]),
      {int(F): round(float(np.median((own[:, 2] * 1e3 / m[:, 0])[m[:, 1] == F]))) for F in np.unique(m[:, 1])})
slow = np.argsort(rel_end)[-4:]
print("slowest:", [(int(i), "C %d F %d R %d longest %d" % (m[i, 0], m[i, 1], m[i, 2], longest[i]), [round(float(x), 1) for x in own[i]]) for i in slow])
