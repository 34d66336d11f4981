import ctypes as C, numpy as np, torch, sys, os
sys.path.insert(0, '.')
import nfst_amd._lib as L
point = sys.argv[1]; mode = sys.argv[2]; B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
prof = C.CDLL(os.path.abspath('scratch/libnfst_prof_%s.so' % point))
for name in L.EXPORTS:
    f = getattr(prof, name); g = getattr(L.lib, name); f.restype = g.restype; f.argtypes = g.argtypes
    setattr(L.lib, name, f)
from nfst_amd import ops, synth
from nfst_amd.lattice import LatticeBatch
lats = synth.bench_batch(B)
lat = LatticeBatch.from_synth(lats).to('cuda')
theta = torch.from_numpy(synth.label_scores(1, 256)).cuda()
out = None
for _ in range(100):
    if mode == "fb": out = ops.forward_backward(lat, theta, out=out)
    else: ops.backward(lat, theta, want_logbeta=False)
torch.cuda.synchronize()
buf = np.zeros(8192, np.uint64)
prof.nfst_debug_read.argtypes = [C.c_void_p]
rc = prof.nfst_debug_read(buf.ctypes.data)
for wave in ([0, 1] if mode == "fb" else [0]):
    tot, accI, accX, polls, n = [int(x) for x in buf[wave * 16: wave * 16 + 5]]
    print("point", point, mode, "B", B, "wave", wave, "tiles", n, "total cycles", tot, "per tile %.1f" % (tot / max(n, 1)),
          "| loop-top to loop-top %.1f" % (accI / max(n - 1, 1)), "| top->%s %.1f" % (point, accX / max(n, 1)), "| polls", polls)
