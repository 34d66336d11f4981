"""Where a tile of the neuralised beta sweep spends its time (profiling build: python -m nfst_amd.build --variant stamps
-DNFST_NEU_STAMPS; NFST_LIB=nfst_amd/lib/variants/libnfst_hip_stamps.so).  Workgroup 0, 100 MHz ticks, summed over tiles."""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nfst_amd import ops, synth, _lib
from nfst_amd.lattice import LatticeBatch

dev = torch.device("cuda:0")
lat = LatticeBatch.from_synth(synth.bench_batch(256), device=dev)
lib = _lib.lib
out = {}
for H in [int(x) for x in (sys.argv[1:] or ["256", "64"])]:
    g = torch.Generator(device="cpu").manual_seed(H)
    lim = (6.0 / (2 * H)) ** 0.5
    prm = [torch.randn(256, H, generator=g), (torch.rand(H, H, generator=g) * 2 - 1) * lim, (torch.rand(H, H, generator=g) * 2 - 1) * lim,
           (torch.rand(1, H, generator=g) * 2 - 1) * (6.0 / (1 + H)) ** 0.5, 0.3 * torch.randn(H, generator=g)]
    prm = [x.to(dev) for x in prm]
    ops.backward_neural(lat, *prm); torch.cuda.synchronize()
    buf = (C.c_ulonglong * 128)()
    lib.nfst_debug_neu_stamps(None, 1)
    ops.backward_neural(lat, *prm); torch.cuda.synchronize()
    lib.nfst_debug_neu_stamps(buf, 0)
    v = list(buf)
    tiles = max(v[5], 1)
    us = lambda x: round(x / 100.0 / tiles, 2)  # microseconds per tile
    names = ["phaseA_own", "wait_barrier_A", "phaseB", "fence", "wait_barrier_B"]
    out[f"H{H}"] = {"tiles": tiles, **{n: [us(v[w * 8 + j]) for w in range(16)] for j, n in enumerate(names)}}
print(json.dumps(out))
