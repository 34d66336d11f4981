"""(Test infrastructure: uses the oracle, hence under tests/; a script, not collected by pytest.)
Randomised check of the posterior sampler: random batches (sizes, widths, degrees up to 60 arcs per state, table
weights, caller scores, K), the three ways k_sample reads a lattice against each other, and every walk checked as an
accepting path with log q = score - log Z.  python tests/fuzz/fuzz_sampling.py [n_batches] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from nfst_amd import ops, synth
from nfst_amd.lattice import LatticeBatch
from oracle import oracle as O

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
worst_q, min_same, t0 = 0.0, 1.0, time.time()
for it in range(n_batches):
    B = int(rng.integers(1, 12))
    V = int(rng.choice([24, 40, 256]))
    weighted = bool(rng.integers(0, 2))
    lats = []
    for i in range(B):
        n = int(rng.choice([4, 5, 6, 9, 17, 40, 90, 200, 450, 900, 2500]))
        md = min(int(rng.choice([8, 24, 60])), (V - 12) // 2)
        for attempt in range(50):
            try:
                lats.append(synth.layered_lattice(int(rng.integers(1, 1 << 30)), n_states=n, avg_degree=min(float(rng.choice([1.5, 3.0, 6.0, 12.0, 20.0])), md / 2),
                                                  vocab=V, width=int(rng.choice([1, 2, 4, 8, 16, 32])), span=int(rng.choice([1, 2, 4, 8])),
                                                  max_degree=md, weighted=weighted))
                break
            except AssertionError:
                continue
    B = len(lats)
    lat = LatticeBatch.from_synth(lats, device=dev)
    theta = synth.label_scores(int(rng.integers(0, 1000)), V, mean=float(rng.choice([-2.3, 0.0, -8.0])), std=float(rng.choice([0.5, 2.0])))
    th = torch.from_numpy(theta)
    asc = rng.normal(0.0, 0.5, size=lat.total_arcs).astype(np.float32) if rng.integers(0, 2) else None
    t = None if asc is None else torch.from_numpy(asc)
    K = int(rng.choice([1, 3, 16, 40, 64, 100]))
    T = int(lat.depth.max()) + 1
    u = torch.from_numpy(rng.random((B, K, T)).astype(np.float32))
    a = ops.sample_paths(lat, th, K, arc_scores=t, max_len=T, uniforms=u)
    b = ops.sample_paths(lat, th, K, arc_scores=t, max_len=T, uniforms=u, want_arcs=False)
    c = ops.sample_paths(lat, th, K, arc_scores=t, seed=it)
    c2 = ops.sample_paths(lat, th, K, arc_scores=t, seed=it)
    torch.cuda.synchronize()
    assert torch.equal(c.paths, c2.paths) and torch.equal(c.arcs, c2.arcs), it
    same = (a.paths == b.paths).all(dim=2)
    min_same = min(min_same, float(same.float().mean()))
    assert float(same.float().mean()) > 0.97, (it, float(same.float().mean()))
    for res in (a, c):
        arcs, lens, logq = res.arcs.cpu().numpy(), res.lengths.cpu().numpy(), res.logq.cpu().numpy()
        for bb in rng.choice(B, size=min(B, 3), replace=False):
            l = lats[bb]
            a0 = int(lat.arc_off[bb])
            sc = theta[l.label].astype(np.float64)
            if l.weight is not None: sc = sc + l.weight
            if asc is not None: sc = sc + asc[a0:a0 + l.n_arcs]
            o = O.forward_backward(l.n_rows, l.src, l.dst, sc)
            for k in range(0, K, max(1, K // 5)):
                p = arcs[bb, k, :lens[bb, k]] - a0
                assert lens[bb, k] > 0 and l.src[p[0]] == 0 and l.dst[p[-1]] == l.n_rows - 1 and np.all(l.dst[p[:-1]] == l.src[p[1:]]), (it, bb, k)
                ref = sc[p].sum() - o["logZ"]
                worst_q = max(worst_q, abs(ref - logq[bb, k]) - 6e-8 * abs(ref))  # (beyond the float32 rounding of log q itself)
    if it % 20 == 19:
        print(f"{it + 1} batches ok, worst log q error beyond float32 rounding {worst_q:.2e}, least agreement of the two staged modes {min_same:.4f}, {time.time() - t0:.0f} s", flush=True)
assert worst_q <= 2e-5, worst_q
print(f"{n_batches} batches ok, worst log q error beyond float32 rounding {worst_q:.2e}, least agreement {min_same:.4f}")
