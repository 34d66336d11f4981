"""The step on deep lattices, general kernels: precise flavour (the default beyond 192 tiles) against the float32 flavour forced, and the
chunked flavour where the packer cut programs: configs[2] SNIPS-shaped batch, the width-4 variant of configs[1], batch size 1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from nfst_amd import ops, synth, _lib
from nfst_amd.lattice import LatticeBatch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import time_op
dev = torch.device("cuda")
def line(name, lat, th):
    out = {"o": None}
    def f(): out["o"] = ops.forward_backward(lat, th, out=out["o"])
    res = {}
    for tag, kw in (("auto", dict(chunked=0)), ("float32", dict(precise=0, chunked=0)), ("chunked", {})):
        with _lib.tuning(**kw):
            out["o"] = None
            res[tag] = time_op(f, 50)
    ck = f"chunked flavour {res['chunked']*1e3:.1f} us" if lat.chunks is not None else "no chunked programs"
    print(f"{name}: lattices {lat.n_lattices} max_tiles {lat.max_tiles}  general kernels {res['auto']*1e3:.1f} us, float32 flavour forced {res['float32']*1e3:.1f} us, {ck}", flush=True)
lats = synth.snips_shaped_batch(64)
th = torch.from_numpy(synth.label_scores(64, lats[0].vocab, mean=-1.5, std=0.8)).to(dev)
line("configs2 snips b64", LatticeBatch.from_synth(lats, device=dev), th)
line("snips b1", LatticeBatch.from_synth(lats[:1], device=dev), th)
th256 = torch.from_numpy(synth.label_scores(1, 256)).to(dev)
line("configs1 width 4", LatticeBatch.from_synth(synth.bench_batch(256, width=4), device=dev), th256)
line("configs1 width 16", LatticeBatch.from_synth(synth.bench_batch(256), device=dev), th256)
