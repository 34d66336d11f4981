"""Chunked flavour against the general kernels on deep, narrow batches: python profiles/tune/chunk_times.py
(ms per launch by events over 50 launches; the SNIPS-shaped batch of BASELINE configs[2], the width-4 variant of configs[1],
one SNIPS-shaped lattice)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from nfst_amd import _lib, ops, synth
from nfst_amd.lattice import LatticeBatch

dev = torch.device("cuda")


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def narrow_batch(n, seed=8000):
    """tagging machines with one to three states per position only (what the chunked flavour is best at)"""
    rng = np.random.default_rng(seed)
    return [synth.layered_lattice(seed + i, n_states=int(rng.integers(400, 1501)), avg_degree=float(rng.choice([3.0, 5.0])), vocab=250,
                                  width=int(rng.choice([1, 2, 3])), span=int(rng.choice([1, 2])), max_degree=40) for i in range(n)]


def main():
    cases = [("snips_b64", synth.snips_shaped_batch(64, vocab=250), 250), ("narrow_b64", narrow_batch(64), 250), ("narrow_b128", narrow_batch(128), 250), ("narrow_b256", narrow_batch(256), 250), ("narrow_b512", narrow_batch(512), 250), ("snips_b1", synth.snips_shaped_batch(1, vocab=250), 250),
             ("snips_b16", synth.snips_shaped_batch(16, vocab=250), 250),
             ("width4_b256", synth.bench_batch(256, width=4), 256), ("width4_b64", synth.bench_batch(64, width=4), 256),
             ("width2_b128", synth.bench_batch(128, width=2), 256)]
    for name, lats, V in cases:
        host = LatticeBatch.from_synth(lats)
        auto = host.build_chunks()
        if not auto:
            host.build_chunks(force=True)
        if host.chunks is None:
            print(name, "no chunked programs"); continue
        lat = host.to(dev)
        m = lat.chunks.meta_host
        theta = torch.from_numpy(synth.label_scores(1, V, mean=-1.5, std=0.8)).to(dev)
        out = {}
        for tag, sw in (("chunked", 1), ("general", 0)):
            with _lib.tuning(chunked=sw):
                o = [None]
                def fb():
                    o[0] = ops.forward_backward(lat, theta, out=o[0])
                out[tag + "_fb"] = timed(fb)
                out[tag + "_bwd"] = timed(lambda: ops.backward(lat, theta, want_logbeta=False))
                z = ops.forward_backward(lat, theta).logz64
                out[tag + "_z"] = z
        err = float((out["chunked_z"] - out["general_z"]).abs().max())
        print(f"{name}: auto={auto} arcs={int(lat.total_arcs)} depth<={int(lat.depth.max())} C {m[:, :, 0].min()}..{m[:, :, 0].max()} F {m[:, :, 1].min()}..{m[:, :, 1].max()} "
              f"lds {lat.chunks._h['lds_bytes']} threads {lat.chunks._h['threads']} | fb chunked {out['chunked_fb']:.1f} us general {out['general_fb']:.1f} us | "
              f"bwd chunked {out['chunked_bwd']:.1f} general {out['general_bwd']:.1f} | max |dlogZ| {err:.2e} flagged {int(lat.chunks.flagged().sum())}", flush=True)


if __name__ == "__main__":
    main()
