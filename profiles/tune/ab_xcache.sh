python - <<'PY'
import time, torch, numpy as np, os
from nfst_amd import ops, synth
from nfst_amd.lattice import LatticeBatch
dev = torch.device("cuda:0")
def t(f, n=300):
    for _ in range(20): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
lats = [synth.layered_lattice(7000 + i, n_states=1000 + (i % 7) * 20, avg_degree=10.0, vocab=256, width=16, span=8) for i in range(256)]
lat = LatticeBatch.from_synth(lats, device=dev); th = torch.from_numpy(synth.label_scores(1, 256)).to(dev)
asc = torch.randn(lat.total_arcs, device=dev) * 0.1
out = ops.forward_backward(lat, th)
print("1k-state lattices, max arcs", lat._h["reserved0"] >> 8, "plain ms", round(t(lambda: ops.forward_backward(lat, th, out=out)), 4))
out2 = ops.forward_backward(lat, th, arc_scores=asc)
for xc in ("1", "0"):
    os.environ["NFST_XCACHE"] = xc
    print("  with scores, NFST_XCACHE =", xc, "ms", round(t(lambda: ops.forward_backward(lat, th, arc_scores=asc, out=out2)), 4))
PY
