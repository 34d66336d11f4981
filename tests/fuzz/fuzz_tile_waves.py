"""(Test infrastructure: uses the oracle, hence under tests/; a script, not collected by pytest.)
Randomised check of the tile-wave kernels (forward-backward, beta only, Viterbi) against the oracle and against the
loader / decoder / sweep pipeline (NFST_TW=0): random batch sizes, lattice sizes, widths, spans, degrees, table weights,
caller scores.  python tests/fuzz/fuzz_tile_waves.py [n_batches] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from nfst_amd import ops, synth
from nfst_amd.lattice import LatticeBatch
from oracle import oracle as O

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
worst = 0.0
t0 = time.time()
for it in range(n_batches):
    B = int(rng.integers(1, 48))
    V = int(rng.choice([24, 40, 256, 700]))
    weighted = bool(rng.integers(0, 2))
    lats = []
    for i in range(B):
        n = int(rng.choice([4, 5, 6, 9, 17, 40, 90, 200, 450, 900]))
        md = min(int(rng.choice([8, 24, 60])), (V - 12) // 2)
        for attempt in range(50):  # (the generator refuses a state whose arcs outnumber the labels: draw again)
            try:
                lats.append(synth.layered_lattice(int(rng.integers(1, 1 << 30)), n_states=n, avg_degree=min(float(rng.choice([1.5, 3.0, 6.0, 12.0])), md / 2),
                                                  vocab=V, width=int(rng.choice([1, 2, 4, 8, 16, 32])), span=int(rng.choice([1, 2, 4, 8])),
                                                  max_degree=md, weighted=weighted))
                break
            except AssertionError:
                continue
    B = len(lats)
    opts = dict(group_mode=int(rng.choice([0, 0, 0, 1, 2])))
    lat = LatticeBatch.from_synth(lats, device=dev, **opts)
    theta = synth.label_scores(int(rng.integers(0, 1000)), V, mean=float(rng.choice([-2.3, 0.0, -8.0])), std=float(rng.choice([0.5, 2.0])))
    th = torch.from_numpy(theta)
    asc = rng.normal(0.0, 0.5, size=lat.total_arcs).astype(np.float32) if rng.integers(0, 2) else None
    t = None if asc is None else torch.from_numpy(asc)
    res = {}
    for tw in ("1", "0"):
        os.environ["NFST_TW"] = tw
        res[tw] = (ops.forward_backward(lat, th, arc_scores=t), ops.backward(lat, th, arc_scores=t), ops.viterbi(lat, th, arc_scores=t))
    torch.cuda.synchronize()
    (fa, ba, va), (fb, bb, vb) = res["1"], res["0"]
    assert torch.equal(fa.logz64, fb.logz64) and torch.equal(fa.logalpha, fb.logalpha) and torch.equal(fa.logbeta, fb.logbeta), it
    assert torch.equal(ba.logz64, bb.logz64) and torch.equal(ba.logbeta, bb.logbeta), it
    # (per-arc extras are several float32 addends per arc: the two Viterbi kernels add them in different orders, so the
    # best value may differ in the last bits and near-ties may resolve differently; without extras everything is bit-exact)
    if asc is None and not weighted:
        assert torch.equal(va.best, vb.best) and torch.equal(va.paths, vb.paths) and torch.equal(va.lengths, vb.lengths), it
    else:
        assert torch.allclose(va.best, vb.best, rtol=1e-5, atol=1e-5) and torch.equal(va.lengths >= 0, vb.lengths >= 0), it
    for b in rng.choice(B, size=min(B, 4), replace=False):
        l = lats[b]
        a0 = int(lat.arc_off[b])
        sc = theta[l.label].astype(np.float64)
        if l.weight is not None: sc = sc + l.weight
        if asc is not None: sc = sc + asc[a0:a0 + l.n_arcs]
        o = O.forward_backward(l.n_rows, l.src, l.dst, sc)
        worst = max(worst, abs(float(fa.logz64[b]) - o["logZ"]))
        err = abs(float(fa.logz64[b]) - o["logZ"])
        if err > 1e-5:
            print("logZ error", err, "batch", it, "lattice", int(b), "rows", l.n_rows, "arcs", l.n_arcs, "depth", int(lat.depth[b]), "logZ", o["logZ"],
                  "weighted", weighted, "scores", asc is not None, flush=True)
        assert err <= (1e-5 if not (weighted or asc is not None) else 3e-5), (it, b)
        assert np.max(np.abs(fa.posterior.cpu().numpy()[a0:a0 + l.n_arcs] - o["posterior"])) <= 1e-5, (it, b)
    if it % 10 == 9:
        print(f"batch {it + 1}/{n_batches} ok, worst |dlogZ| {worst:.2e}, {time.time() - t0:.0f} s", flush=True)
print("fuzz ok", n_batches, "batches, worst |dlogZ|", worst)
