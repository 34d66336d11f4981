"""Throughput of the secondary kernels of the path (not the BASELINE metric)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nfst_amd import ops, synth
from nfst_amd.lattice import LatticeBatch
dev = torch.device('cuda')
def timeit(f, n=50, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
out = {}
# path_logprob: N x T x V float32 read once
N, T, V = 2048, 300, 300
scores = torch.randn(N, T, V, device=dev)
marks = torch.randint(3, V, (N, T), device=dev); marks[:, -1] = 2
t = timeit(lambda: ops.path_logprob(scores, marks, max_length=400))
out["path_logprob"] = {"shape": [N, T, V], "ms": t * 1e3, "GB/s": N * T * V * 4 / t / 1e9}
# its backward: the rows are recomputed, N x T x V float32 read + as many written
for tag, kw in (("eval", dict()), ("smoothing", dict(smoothing=0.1))):
    sg = scores.clone().requires_grad_(True)
    val = ops.path_logprob(sg, marks, max_length=400, **kw)
    gout = torch.randn(N, device=dev)
    t = timeit(lambda: torch.autograd.grad(val, sg, gout, retain_graph=True), n=20, w=3)
    out[f"path_logprob_backward_{tag}"] = {"shape": [N, T, V], "ms": t * 1e3, "GB/s_read_plus_written": 2 * N * T * V * 4 / t / 1e9}
    del sg, val
for V2 in (128, 256, 512, 1000):
    s2 = torch.randn(1024, 128, V2, device=dev, requires_grad=True)
    m2 = torch.randint(3, V2, (1024, 128), device=dev)
    v2 = ops.path_logprob(s2, m2, max_length=400)
    g2 = torch.randn(1024, device=dev)
    t = timeit(lambda: torch.autograd.grad(v2, s2, g2, retain_graph=True), n=20, w=3)
    out[f"path_logprob_backward_V{V2}"] = {"shape": [1024, 128, V2], "ms": t * 1e3, "GB/s_read_plus_written": 2 * 1024 * 128 * V2 * 4 / t / 1e9}
    del s2, v2
del scores
lats = synth.bench_batch(256)
lat = LatticeBatch.from_synth(lats, device=dev)
theta = torch.from_numpy(synth.label_scores(1, 256)).to(dev)
arcs = int(lat.n_dp_arcs.sum())
t = timeit(lambda: ops.backward(lat, theta, want_logbeta=True))
out["backward"] = {"ms": t * 1e3, "arcs/s": arcs / t, "GB/s_algorithmic": lat.algorithmic_bytes("backward") / t / 1e9}
t = timeit(lambda: ops.viterbi(lat, theta), n=10, w=2)
out["viterbi"] = {"ms": t * 1e3, "arcs/s": arcs / t}
beta = ops.backward(lat, theta, want_logbeta=False, want_me=True)
for K in (16, 64):
    t = timeit(lambda: ops.sample_paths(lat, theta, K, seed=1, beta=beta), n=10, w=2)
    out[f"sample_paths_K{K}"] = {"ms": t * 1e3, "walks/s": 256 * K / t, "arc-steps/s": 256 * K * 130 / t}
st = torch.zeros(256 * 64, dtype=torch.int64, device=dev); lb = torch.ones(256 * 64, dtype=torch.int64, device=dev)
t = timeit(lambda: ops.step(lat, st, lb, k=64))
out["step_K64"] = {"ms": t * 1e3, "walkers/s": 256 * 64 / t}
t = timeit(lambda: ops.emission_mask(lat, st, k=64, inp=lb))
out["emission_mask_K64"] = {"ms": t * 1e3, "GB/s_written": 256 * 64 * 256 * 4 / t / 1e9}
sc64 = torch.randn(256 * 64, 256, device=dev); u64 = torch.rand(256 * 64, device=dev)
t = timeit(lambda: ops.proposal_step(lat, st, sc64, k=64, inp=lb, uniforms=u64))
out["proposal_step_K64"] = {"ms": t * 1e3, "walkers/s": 256 * 64 / t, "note": "incl. the Python wrapper (4 output allocations)"}
# the sampler loop (Sampler.stateful_sample, samplers.py:243-297): T ~ 300 steps, N = 2048 walkers, a network that costs nothing;
# sync_every = 1 is the reference's protocol (all_reached_eos looked at, i.e. a device synchronisation, at every step)
from nfst_amd.samplers import ProposalSampler
from nfst_amd.scorers import LatticeScorer
lats_s = [synth.layered_lattice(500 + i, n_states=1150, avg_degree=4.0, vocab=64, width=4, span=1, max_degree=12) for i in range(32)]
sc = LatticeScorer(64, pad=0, bos=1, eos=2, max_length=300).to(dev)
em_s, tr_s = synth.collate_dense([l.dense() for l in lats_s])
zero_logits = torch.zeros(32 * 64, 64, device=dev)
for every in (1, 8):
    sp = ProposalSampler(sc, lambda hx, inp: (hx, zero_logits), sync_every=every)
    sp.set_masks(transition=torch.from_numpy(tr_s), emission=torch.from_numpy(em_s))
    sp.set_k(64)
    with torch.no_grad():
        sp.stateful_sample(32 * 64)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            lq, smp, _ = sp.stateful_sample(32 * 64)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 3
    out[f"proposal_sampler_T{smp.shape[1]}_N2048_sync_every_{every}"] = {"ms": t * 1e3, "us_per_step": t / (smp.shape[1] + 1) * 1e6}
t = timeit(lambda: ops.gather_label_scores(lat, theta))
out["gather_label_scores"] = {"ms": t * 1e3, "GB/s": lat.total_arcs * 8 / t / 1e9}
# neuralised beta: hid_dim of the reference's configs (conf/train/*.yaml: 256) and a small one
for H in (256, 64, 8):
    g = torch.Generator(device="cpu").manual_seed(H)
    lim = (6.0 / (2 * H)) ** 0.5
    prm = [torch.randn(256, H, generator=g), (torch.rand(H, H, generator=g) * 2 - 1) * lim, (torch.rand(H, H, generator=g) * 2 - 1) * lim,
           (torch.rand(1, H, generator=g) * 2 - 1) * (6.0 / (1 + H)) ** 0.5, 0.3 * torch.randn(H, generator=g)]
    prm = [x.to(dev) for x in prm]
    t = timeit(lambda: ops.backward_neural(lat, *prm), n=5, w=1)
    states = int(lat.meta_host[:, 9].sum())
    out[f"backward_neural_H{H}"] = {"ms": t * 1e3, "arcs/s": arcs / t,
                                   "GFLOP/s": (2.0 * states * H * H + 4.0 * arcs * H) / t / 1e9}
    # its gradient (nfst_backward_neural_grad + two library GEMMs), forward pass not included
    pg = [x.clone().requires_grad_(True) for x in prm]
    r = ops.backward_neural(lat, *pg)
    fin = torch.isfinite(r.log_beta)
    loss = r.log_beta[fin].sum()
    t = timeit(lambda: torch.autograd.grad(loss, pg, retain_graph=True), n=5, w=1)
    out[f"backward_neural_grad_H{H}"] = {"ms": t * 1e3, "arcs/s": arcs / t}
    del pg, r, loss
asc = torch.randn(lat.total_arcs, device=dev) * 0.1
t = timeit(lambda: ops.viterbi(lat, theta, arc_scores=asc), n=10, w=2)
out["viterbi_with_arc_scores"] = {"ms": t * 1e3, "arcs/s": arcs / t}
t = timeit(lambda: ops.backward(lat, theta, arc_scores=asc, want_logbeta=True))
out["backward_with_arc_scores"] = {"ms": t * 1e3, "arcs/s": arcs / t}
print(json.dumps(out, indent=1))
