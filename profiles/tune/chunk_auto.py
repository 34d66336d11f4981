"""Does the packer's cost model pick the faster flavour?  SNIPS-shaped and narrow batches of 1 .. 32 lattices: forward-backward
on the chunked flavour (forced) and on the general kernels, and what LatticeBatch.to() would have chosen."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from nfst_amd import _lib, ops, synth
from nfst_amd.lattice import LatticeBatch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from chunk_times import timed, narrow_batch  # noqa: E402

dev = torch.device("cuda")
right = total = 0
for kind in ("snips", "narrow"):
    for B in (1, 2, 4, 8, 32):
        for seed in (0, 1, 2):
            lats = synth.snips_shaped_batch(B, vocab=250, first_seed=3000 + 500 * seed) if kind == "snips" else narrow_batch(B, seed=8000 + 500 * seed)
            if kind == "snips":  # (the generator draws shapes from a fixed stream: vary the batch by skipping)
                lats = synth.snips_shaped_batch(B + 8 * seed, vocab=250)[8 * seed:]
            host = LatticeBatch.from_synth(lats)
            auto = host.build_chunks()
            if not auto and not host.build_chunks(force=True):
                print(kind, B, seed, "cannot be cut"); continue
            lat = host.to(dev)
            theta = torch.from_numpy(synth.label_scores(1, 250, mean=-1.5, std=0.8)).to(dev)
            t = {}
            for tag, sw in (("chunked", 1), ("general", 0)):
                with _lib.tuning(chunked=sw):
                    o = [None]
                    def fb():
                        o[0] = ops.forward_backward(lat, theta, out=o[0])
                    t[tag] = timed(fb, 30)
            best = "chunked" if t["chunked"] < t["general"] else "general"
            ok = (best == "chunked") == auto or abs(t["chunked"] - t["general"]) < 0.1 * t["general"]
            right += ok; total += 1
            print(f"{kind} B={B} seed={seed}: depth<={int(lat.depth.max())} F<={int(lat.chunks.meta_host[:, :, 1].max())} chunked {t['chunked']:.1f} us general {t['general']:.1f} us auto={'chunked' if auto else 'general'} {'ok' if ok else 'WRONG'}", flush=True)
print(f"{right} of {total} choices right (or within 10 %)")
