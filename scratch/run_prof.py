import ctypes as C, numpy as np, torch, sys, os
sys.path.insert(0, '.')
import nfst_amd._lib as L
# swap in the profiling library
prof = C.CDLL(os.path.abspath('scratch/libnfst_prof.so'))
for name in L.EXPORTS:
    f = getattr(prof, name); g = getattr(L.lib, name); f.restype = g.restype; f.argtypes = g.argtypes
    setattr(L.lib, name, f)
from nfst_amd import ops, synth
from nfst_amd.lattice import LatticeBatch
mode = sys.argv[1] if len(sys.argv) > 1 else "fb"
lats = synth.bench_batch(256)
lat = LatticeBatch.from_synth(lats).to('cuda')
theta = torch.from_numpy(synth.label_scores(1,256)).cuda()
for _ in range(5):
    r = ops.forward_backward(lat, theta) if mode == "fb" else ops.backward(lat, theta)
torch.cuda.synchronize()
buf = np.zeros(8192, np.uint64)
prof.nfst_debug_read.argtypes = [C.c_void_p]
rc = prof.nfst_debug_read(buf.ctypes.data); print("rc", rc)
for wave in ([0,1,4,5] if mode == "fb" else [0,1,2,3]):
    d = buf[wave*512: wave*512 + 32*8].reshape(32,8).astype(np.int64)
    seg = np.diff(d[:, :7], axis=1)   # 0->1 setup, 1->2 loop, 2->3 reduce, 3->4 write, 4->5 wait, 5->6 barrier
    per_step = np.diff(d[:,0])
    print("wave", wave, "median cycles/step", np.median(per_step), "segments median", np.median(seg, axis=0), "mean", seg.mean(axis=0).round(0))
