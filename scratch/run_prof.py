import ctypes as C, numpy as np, torch, sys, os
sys.path.insert(0, '.')
import nfst_amd._lib as L
prof = C.CDLL(os.path.abspath('scratch/libnfst_prof.so'))
for name in L.EXPORTS:
    f = getattr(prof, name); g = getattr(L.lib, name); f.restype = g.restype; f.argtypes = g.argtypes
    setattr(L.lib, name, f)
from nfst_amd import ops, synth
from nfst_amd.lattice import LatticeBatch
mode = sys.argv[1]; W = int(sys.argv[2]); width = int(sys.argv[3]) if len(sys.argv) > 3 else 16
lats = synth.bench_batch(256, width=width)
lat = LatticeBatch.from_synth(lats, sweep_waves=W).to('cuda')
theta = torch.from_numpy(synth.label_scores(1,256)).cuda()
for _ in range(5):
    r = ops.forward_backward(lat, theta) if mode == "fb" else ops.backward(lat, theta)
torch.cuda.synchronize()
buf = np.zeros(8192, np.uint64)
prof.nfst_debug_read.argtypes = [C.c_void_p]
rc = prof.nfst_debug_read(buf.ctypes.data)
waves = range(2*W) if mode == "fb" else range(W)
print("mode", mode, "W", W, "width", width, " segments: 0 advance+hdr | 1 gathers+prefetch issue | 2 sum | 3 remainder | 4 reduce+write | 5 rotate/barrier")
for wave in waves:
    d = buf[wave*512: wave*512 + 32*8].reshape(32,8).astype(np.int64)
    seg = np.diff(d[:, :7], axis=1)
    per_step = np.diff(d[:,0])
    print("wave", wave, "median cycles/step", np.median(per_step), "segments median", np.median(seg, axis=0))
