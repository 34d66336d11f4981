"""Neuralised beta sweep on the BASELINE batch for several hidden sizes (kernel time by HIP events around the op).
NFST_NEU_NO_SMALL=1 selects the two-phase kernel for H <= 64 (A/B)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nfst_amd import ops, synth
from nfst_amd.lattice import LatticeBatch

dev = torch.device("cuda:0")
lat = LatticeBatch.from_synth(synth.bench_batch(256), device=dev)
out = {}
for H in [int(x) for x in (sys.argv[1:] or ["8", "16", "32", "64", "128", "256"])]:
    g = torch.Generator(device="cpu").manual_seed(H)
    lim = (6.0 / (2 * H)) ** 0.5
    prm = [torch.randn(256, H, generator=g), (torch.rand(H, H, generator=g) * 2 - 1) * lim, (torch.rand(H, H, generator=g) * 2 - 1) * lim,
           (torch.rand(1, H, generator=g) * 2 - 1) * (6.0 / (1 + H)) ** 0.5, 0.3 * torch.randn(H, generator=g)]
    prm = [x.to(dev) for x in prm]
    for _ in range(2): r = ops.backward_neural(lat, *prm)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): r = ops.backward_neural(lat, *prm)
    e1.record(); torch.cuda.synchronize()
    out[f"H{H}"] = round(e0.elapsed_time(e1) / 5, 4)
    out[f"H{H}_sum"] = float(r.log_beta[torch.isfinite(r.log_beta)].double().sum())
print(json.dumps(out))
