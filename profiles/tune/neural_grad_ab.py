"""Gradient of the neuralised beta sweep on the BASELINE batch (nfst_backward_neural_grad + the host-side GEMMs),
forward pass not included; HIP events around torch.autograd.grad."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nfst_amd import ops, synth
from nfst_amd.lattice import LatticeBatch

dev = torch.device("cuda:0")
lat = LatticeBatch.from_synth(synth.bench_batch(256), device=dev)
out = {}
for H in [int(x) for x in (sys.argv[1:] or ["8", "64", "256"])]:
    g = torch.Generator(device="cpu").manual_seed(H)
    lim = (6.0 / (2 * H)) ** 0.5
    prm = [torch.randn(256, H, generator=g), (torch.rand(H, H, generator=g) * 2 - 1) * lim, (torch.rand(H, H, generator=g) * 2 - 1) * lim,
           (torch.rand(1, H, generator=g) * 2 - 1) * (6.0 / (1 + H)) ** 0.5, 0.3 * torch.randn(H, generator=g)]
    pg = [x.to(dev).requires_grad_(True) for x in prm]
    r = ops.backward_neural(lat, *pg)
    loss = r.log_beta[torch.isfinite(r.log_beta)].sum()
    for _ in range(2): gr = torch.autograd.grad(loss, pg, retain_graph=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): gr = torch.autograd.grad(loss, pg, retain_graph=True)
    e1.record(); torch.cuda.synchronize()
    out[f"H{H}"] = round(e0.elapsed_time(e1) / 5, 4)
    out[f"H{H}_norms"] = [float(x.double().norm()) for x in gr]
print(json.dumps(out))
