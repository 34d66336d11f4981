"""Neuralised beta, phase B on three bfloat16 parts (default, hid a multiple of 64) against float32 MFMAs (tuning neu_bf16 = 0):
kernel time of the forward sweep and of the gradient op on the BASELINE batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from nfst_amd import ops, synth, _lib
from nfst_amd.lattice import LatticeBatch
from bench import time_op
dev = torch.device("cuda")
lat = LatticeBatch.from_synth(synth.bench_batch(256), device=dev)
for H in (64, 128, 256, 512):
    g = torch.Generator(device="cpu").manual_seed(H)
    lim = (6.0 / (2 * H)) ** 0.5
    prm = [torch.randn(256, H, generator=g), (torch.rand(H, H, generator=g) * 2 - 1) * lim, (torch.rand(H, H, generator=g) * 2 - 1) * lim,
           (torch.rand(1, H, generator=g) * 2 - 1) * (6.0 / (1 + H)) ** 0.5, 0.3 * torch.randn(H, generator=g)]
    prm = [x.to(dev) for x in prm]
    res = {}
    for bf in (1, 0):
        with _lib.tuning(neu_bf16=bf):
            with torch.no_grad():
                fwd = time_op(lambda: ops.backward_neural(lat, *prm), 5, 2)
                ref = ops.backward_neural(lat, *prm)
            pg = [x.clone().requires_grad_(True) for x in prm]
            r = ops.backward_neural(lat, *pg)
            loss = r.log_beta[torch.isfinite(r.log_beta)].sum()
            grad = time_op(lambda: torch.autograd.grad(loss, pg, retain_graph=True), 5, 2)
            gr = torch.autograd.grad(loss, pg, retain_graph=True)
        res[bf] = (fwd, grad, ref.log_beta.clone(), [x.clone() for x in gr])
    fin = torch.isfinite(res[1][2])
    dlb = float((res[1][2][fin] - res[0][2][fin]).abs().max())
    dg = max(float(((a - b).abs().max() / (b.abs().max() + 1e-30))) for a, b in zip(res[1][3], res[0][3]))
    print(f"H={H}: forward {res[1][0]:.3f} ms (float32 MFMA {res[0][0]:.3f}), gradient op {res[1][1]:.3f} ms ({res[0][1]:.3f}); "
          f"max |d log beta| {dlb:.2e}, max relative gradient difference {dg:.2e}", flush=True)
