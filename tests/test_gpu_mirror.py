"""The host-side mirrors of the reference's Python interface (LatticeScorer, Sampler,
Estimators, JointProb) driving the HIP engine -- parity with the golden fixtures and
the oracle.  Needs the MI355X."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O
from nfst_amd import io, synth
from nfst_amd.estimators import Estimators
from nfst_amd.joint import JointProb
from nfst_amd.lattice import LatticeBatch
from nfst_amd.samplers import Sampler
from nfst_amd.scorers import LatticeScorer

pytestmark = pytest.mark.gpu
PAD, BOS, EOS = synth.PAD, synth.BOS, synth.EOS


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def test_scorer_mirrors_reference_methods(dev, golden_dir):
    d = load(golden_dir, "gather")
    K, maxlen = int(d["K"]), int(d["max_length"])
    V = d["pad7_emission"].shape[2]
    sc = LatticeScorer(V, pad=PAD, bos=BOS, eos=EOS, max_length=maxlen).to(dev)
    sc.set_masks(emission=torch.from_numpy(d["pad7_emission"]), transition=torch.from_numpy(d["pad7_transition"]))
    sc.set_k(K)
    for r in range(0, 40, 7):
        st = torch.from_numpy(d["pad7_states"][r]).to(dev)
        lb = torch.from_numpy(d["pad7_labels"][r]).to(dev)
        assert np.array_equal(sc.update_fsa_state(lb, st).cpu().numpy(), d["pad7_next"][r])
        assert np.array_equal(sc.mask_out_invalid(lb, {"state": st, "length": 5}).cpu().numpy(), d["pad7_mask_len5"][r])
        assert np.array_equal(sc.mask_out_invalid(lb, {"state": st, "length": 21}).cpu().numpy(), d["pad7_mask_len21"][r])
    with pytest.raises(AssertionError):
        sc.set_masks(emission=torch.zeros(3, 4), transition=torch.zeros(3, 4, dtype=torch.long))  # scorers.py:878


def test_scorer_compute_beta_matches_reference(dev, golden_dir):
    d = load(golden_dir, "beta_edit")
    V = d["emission"].shape[1]
    sc = LatticeScorer(V, theta=torch.from_numpy(d["theta"])).to(dev)
    sc.set_masks(emission=torch.from_numpy(d["emission"])[None], transition=torch.from_numpy(d["transition"])[None])
    sc.set_k(3)
    beta = sc.compute_beta().cpu().numpy()
    assert beta.shape == (3, d["emission"].shape[0])
    reach = beta[0] > 0
    assert np.allclose(beta[0][reach], d["beta_per_sample"][reach], rtol=2e-5)
    assert np.allclose(beta[0][reach], d["beta_parallel"][reach], rtol=2e-5)  # no parallel arcs in this lattice
    assert np.array_equal(beta[0], beta[2])


def test_sampler_and_iwae_contract(dev, golden_dir):
    d = load(golden_dir, "iwae")
    V = d["emission"].shape[2]
    theta = torch.from_numpy(d["theta"])
    model = LatticeScorer(V, theta=theta, max_length=48).to(dev)
    sampler = Sampler(model)
    B, K = d["emission"].shape[0], 16
    sampler.set_masks(transition=torch.from_numpy(d["transition"]), emission=torch.from_numpy(d["emission"]))
    sampler.set_k(K)
    log_q, samples = sampler.sample(B * K)
    assert log_q.shape == (B * K,) and samples.dim() == 2 and samples.shape[0] == B * K and samples.dtype == torch.int64
    s = samples.cpu().numpy()
    assert np.any(s[:, -1] != PAD)  # the trailing all-pad column is dropped (samplers.py:304-307)
    # every sample is an accepting path of its lattice (oracle forced walk), log q = score - log Z
    for b in range(B):
        src, label, dst, _ = O.dense_to_arcs(d["emission"][b], d["transition"][b])
        n_rows = d["transition"][b].shape[0]
        sc = d["theta"][label].astype(np.float64)
        r = O.forward_backward(n_rows, src, dst, sc)
        marks = np.concatenate([np.full((K, 1), BOS), s[b * K:(b + 1) * K]], axis=1).astype(np.int32)
        tot, end = O.score_paths(n_rows, src, label, dst, sc, marks)
        sink = int(model._lat().sink[b])
        assert np.all(end == sink)
        assert np.max(np.abs(tot - r["logZ"] - log_q.cpu().numpy()[b * K:(b + 1) * K])) <= 2e-5
    # stripping_pad == the reference's loop (oracle restatement) on the reference's own samples
    ref = d["samples"].reshape(-1, d["samples"].shape[2])
    got = sampler.stripping_pad(torch.from_numpy(ref).to(dev)).cpu().numpy()
    assert np.array_equal(got, O.stripping_pad(ref, PAD))
    # forced scoring of the samples reproduces log q
    (lq2,) = sampler.sample(B * K, to_evaluate=samples)
    assert torch.max(torch.abs(lq2 - log_q)) <= 2e-5
    # IWAE: same 4-tuple as estimatros.py:33-44; exact posterior proposal => zero variance
    lm, lq, smp, log_w = Estimators.iwae(sampler, model, B, K, 0, None)
    assert lm.shape == (B,) and lq.shape == (B, K) and smp.shape[:2] == (B, K) and log_w.shape == (B, K)
    z = Estimators.exact(sampler).detach()
    bos_score = float(d["theta"][BOS])  # the sampler's implicit bos arc is part of Z but not of the samples
    assert torch.max(torch.abs(log_w - (z[:, None] - bos_score))) <= 5e-5
    assert torch.max(torch.abs(lm - (z - bos_score))) <= 5e-5


def test_joint_prob_forward_and_decode(dev, tmp_path):
    V = 48
    lats = [synth.edit_lattice([10, 11, 12, 13], [20, 21, 22], vocab=V, seed=3),
            synth.layered_lattice(2, n_states=80, avg_degree=4.0, vocab=V, width=5, span=3)]
    theta = synth.label_scores(4, V)
    jp = JointProb(V, pad=PAD, bos=BOS, eos=EOS, k=8, theta=torch.from_numpy(theta)).to(dev)
    batch = io.collate([(l.dense()[0], l.dense()[1], l.dense()[0], l.dense()[1], np.arange(3), np.arange(4)) for l in lats],
                       pad=PAD)
    tb = tuple(torch.from_numpy(x) for x in batch)
    num, den = jp(*tb)
    assert den.abs().max() == 0
    for b, l in enumerate(lats):
        o = O.forward_backward(l.n_rows, l.src, l.dst, theta[l.label].astype(np.float64))
        assert abs(float(num[b]) - o["logZ"]) <= 1e-5
    # training signal: d(-mean log Z)/d theta = -expected mark counts / B
    (-num.mean()).backward()
    ref = np.zeros(V)
    for l in lats:
        o = O.forward_backward(l.n_rows, l.src, l.dst, theta[l.label].astype(np.float64))
        ref -= np.bincount(l.label, weights=o["posterior"], minlength=V) / len(lats)
    assert np.max(np.abs(jp.tilde_p.theta.grad.cpu().numpy() - ref)) <= 1e-4
    # decode_from_npz (lightning.py:647-658) on a record written in the reference's format
    em, tr = lats[0].dense()
    path = os.path.join(tmp_path, "x.npz")
    io.save_fsa_npz(path, (em, tr), (em, tr), gs=[1, 2], ps=[3])
    prob, mark = jp.decode_from_npz(path, V, PAD)
    best, vpath, _ = O.viterbi(lats[0].n_rows, lats[0].src, lats[0].label, lats[0].dst, theta[lats[0].label], 512)
    o = O.forward_backward(lats[0].n_rows, lats[0].src, lats[0].dst, theta[lats[0].label].astype(np.float64))
    assert abs(prob - o["logZ"]) <= 1e-5
    assert np.array_equal(mark.cpu().numpy(), vpath[1:])
    # the IWAE path of the reference gives the same number with the exact proposal
    jp2 = JointProb(V, pad=PAD, bos=BOS, eos=EOS, k=8, theta=torch.from_numpy(theta), exact=False).to(dev)
    num2, _ = jp2(*tb)
    assert torch.max(torch.abs(num2 - (num.detach() - float(theta[BOS])))) <= 5e-5


def test_neural_beta_scorer_matches_reference(dev, golden_dir):
    """NeuralBetaScorer.compute_beta == FSAGRUScorer.compute_beta_per_sample with the reference's own
    parameters (Wh != 0), called the way Sampler.stateful_sample calls it (samplers.py:196-198)."""
    from nfst_amd.scorers import NeuralBetaScorer
    name = "neural_layered40_h16"
    with np.load(os.path.join(golden_dir, "beta_neural.npz")) as g:
        c = {k[len(name) + 1:]: g[k] for k in g.files if k.startswith(name + "_")}
    V, H = c["emb"].shape
    sc = NeuralBetaScorer(H, V, pad=PAD, bos=BOS, eos=EOS).to(dev)
    with torch.no_grad():
        sc.embeddings.weight.copy_(torch.from_numpy(c["emb"]))
        for k, p in (("Wx", sc.Wx), ("Wh", sc.Wh), ("W", sc.W), ("bias", sc.beta_bias)):
            p.copy_(torch.from_numpy(c[k]))
    tr = c["transition"]
    em = tr != 0
    em[-1, PAD] = True  # the sink's pad self loop (scorers.py:1013-1016)
    tr = tr.copy(); tr[-1, PAD] = tr.shape[0] - 1
    sc.set_masks(emission=torch.from_numpy(em)[None], transition=torch.from_numpy(tr)[None])
    sc.set_k(3)
    beta = sc.compute_beta().detach().cpu().numpy()
    assert beta.shape == (3, tr.shape[0]) and np.array_equal(beta[0], beta[2])
    np.testing.assert_allclose(beta[0], c["beta_per_sample"], rtol=5e-5)
    _, bhat = sc.compute_beta_hat()
    bhat = bhat.detach()
    assert bhat.shape == (3, tr.shape[0], H) and float(bhat.abs().max()) <= 1.0
    # trainable as in tune_proposal (lightning.py:339-406): the scorer's parameters get gradients through
    # compute_beta (they match float64 autograd over the restatement in test_gpu_parity.py)
    lb = sc.compute_log_beta()
    lb[0][torch.isfinite(lb[0])].sum().backward()
    for p in (sc.Wh, sc.Wx, sc.W, sc.beta_bias, sc.embeddings.weight):
        assert p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0


def test_proposal_sampler_replays_reference_sampler(dev, golden_dir):
    """ProposalSampler.stateful_sample (one fused launch per step) with the FSAMaskScorer proposal
    (uniform logits): forced along the reference sampler's own samples it returns the reference's
    log q; sampling freely, every sample is an accepting path whose log q is -sum log(out-degree)."""
    from nfst_amd.samplers import ProposalSampler
    from nfst_amd import ops
    d = load(golden_dir, "sampler")
    K, max_length = int(d["K"]), int(d["max_length"])
    B, _, V = d["emission"].shape
    sc = LatticeScorer(V, pad=PAD, bos=BOS, eos=EOS, max_length=max_length).to(dev)
    steps = []

    def score_fn(hx, inp):
        steps.append(1)
        return hx, torch.zeros(inp.shape[0], V, device=dev)

    sp = ProposalSampler(sc, score_fn)
    sp.set_masks(transition=torch.from_numpy(d["transition"]), emission=torch.from_numpy(d["emission"]))
    sp.set_k(K)
    N = B * K
    log_q, hx = sp.stateful_sample(N, to_evaluate=torch.from_numpy(d["samples"]))
    assert np.max(np.abs(log_q.cpu().numpy() - d["log_q"])) < 1e-5 and hx is None
    steps.clear()
    g = torch.Generator().manual_seed(3)
    u = torch.rand(max_length + 1, N, generator=g)
    log_q, samples, _ = sp.stateful_sample(N, uniforms=u)
    # (the loop reads the device-side "every walker has ended" counters every 8 steps: the network may have been asked
    # for up to 7 steps past the last one, whose results are dropped; the reference synchronises at every step)
    assert samples.shape[0] == N and samples.shape[1] + 1 <= len(steps) <= samples.shape[1] + 8
    assert len(steps) % 8 == 0 or len(steps) == max_length + 1
    arcs = [O.dense_to_arcs(d["emission"][b], d["transition"][b]) for b in range(B)]
    s_np, q_np = samples.cpu().numpy(), log_q.cpu().numpy()
    for n in range(N):
        src, lab, dst = arcs[n // K][:3]
        state, ref = int(dst[(src == 0) & (lab == BOS)][0]), 0.0
        for t in range(s_np.shape[1]):
            out = (src == state) & (dst != state)
            if s_np[n, t] == PAD:
                assert not out.any()  # only in the sink
                continue
            legal = out & (lab == s_np[n, t])
            assert legal.sum() == 1
            ref -= np.log(out.sum())
            state = int(dst[legal][0])
        assert state == d["transition"].shape[1] - 1 or not ((src == state) & (dst != state)).any()
        assert abs(q_np[n] - ref) < 1e-5


@pytest.mark.parametrize("tag", ["a", "b"])
def test_proposal_sampler_replays_reference_use_beta_sampler(dev, golden_dir, tag):
    """The reference's learned proposal FSAGRUScorer(use_beta=True) under Sampler.stateful_sample
    (fixture sampler_beta.npz: recorded per-step prefix scores of the recurrent cell, beta of
    compute_beta(), insertion and length penalties, temperature): ProposalSampler, forced along the
    reference's samples, returns the reference's log q -- with the beta gather out of the state before
    the previous symbol was consumed, as the reference does (scorers.py:584-590 vs 679)."""
    from nfst_amd.samplers import ProposalSampler
    d = load(golden_dir, "sampler_beta")
    K = int(d["K"])
    ins_thr, ins_pen, len_thr, len_pen, temperature, max_length, ins_mark = d[f"{tag}_cfg"]
    B, R, V = d["emission"].shape
    N = B * K
    sc = LatticeScorer(V, pad=PAD, bos=BOS, eos=EOS, max_length=int(max_length)).to(dev)
    pre = torch.from_numpy(d[f"{tag}_prefix_scores"]).to(dev)
    step = {"t": 0}

    def score_fn(hx, inp):
        t = step["t"]
        step["t"] += 1
        return hx, pre[t]

    pen = dict(insertion_mark=int(ins_mark), insert_threshold=int(ins_thr), insert_penalty=float(ins_pen),
               length_threshold=int(len_thr), length_penalty=float(len_pen))
    sp = ProposalSampler(sc, score_fn, penalties=pen)
    sp.set_masks(transition=torch.from_numpy(d["transition"]), emission=torch.from_numpy(d["emission"]))
    sp.set_k(K)
    beta = torch.from_numpy(d[f"{tag}_beta"][::K]).to(dev).reshape(-1)  # [B * (S+1)] row-indexed, probability domain
    log_q, hx = sp.stateful_sample(N, to_evaluate=torch.from_numpy(d[f"{tag}_samples"]), values=beta,
                                   temperature=float(temperature))
    assert np.max(np.abs(log_q.cpu().numpy() - d[f"{tag}_log_q"])) < 1e-5
    # the other order (beta gathered from the advanced state) does not reproduce the reference
    step["t"] = 0
    sp2 = ProposalSampler(sc, score_fn, penalties=pen, beta_from_previous_state=False)
    log_q2, _ = sp2.stateful_sample(N, to_evaluate=torch.from_numpy(d[f"{tag}_samples"]), values=beta,
                                    temperature=float(temperature))
    assert np.max(np.abs(log_q2.cpu().numpy() - d[f"{tag}_log_q"])) > 1e-3
    # sampling freely with the penalties on: every sample ends, log q is finite
    step["t"] = 0
    u = torch.rand(int(max_length) + 1, N, generator=torch.Generator().manual_seed(5))
    pre_long = torch.zeros(int(max_length) + 1, N, V, device=dev)
    pre_long[: pre.shape[0]] = pre

    def score_long(hx, inp):
        t = step["t"]
        step["t"] += 1
        return hx, pre_long[t]

    sp3 = ProposalSampler(sc, score_long, penalties=pen)
    lq3, samples3, _ = sp3.stateful_sample(N, values=beta, temperature=float(temperature), uniforms=u)
    assert torch.isfinite(lq3).all() and samples3.shape[0] == N


def test_proposal_step_autograd(dev):
    """log q and log z of the fused step are differentiable in the proposal's scores and in the
    gathered values; checked against torch.autograd (float64, CPU) on the dense restatement of the
    step: logits = ((scores + values[next state]) * padmask + masks) / T."""
    from nfst_amd import ops
    from oracle import oracle as O
    V, K = 48, 6
    lats = [synth.layered_lattice(700 + i, n_states=40 + 25 * i, avg_degree=5.0, vocab=V, width=4, span=3) for i in range(3)]
    em, tr = synth.collate_dense([l.dense() for l in lats])
    lat = LatticeBatch.from_dense(em, tr, device=dev)
    R = em.shape[1]
    N = len(lats) * K
    rng = np.random.default_rng(9)
    em_k, tr_k = O.expand_k(em, K), O.expand_k(tr, K)
    vstate = np.zeros(N, np.int64); state = np.zeros(N, np.int64); inp = np.zeros(N, np.int64)
    for n in range(N):
        l = lats[n // K]
        a = int(rng.integers(0, l.n_arcs - 1))
        while l.src[a] == l.dst[a]:
            a = int(rng.integers(0, l.n_arcs - 1))
        vstate[n], inp[n], state[n] = l.src[a], l.label[a], l.dst[a]
    scores = rng.normal(0, 1.2, size=(N, V)).astype(np.float32)
    values = rng.normal(0, 0.6, size=lat.total_rows).astype(np.float32)
    u = rng.random(N).astype(np.float32)
    gq, gz = rng.normal(size=N).astype(np.float32), rng.normal(size=N).astype(np.float32)
    for T, own in ((1.0, True), (0.7, False)):
        sc = torch.from_numpy(scores).to(dev).requires_grad_(True)
        vl = torch.from_numpy(values).to(dev).requires_grad_(True)
        r = ops.proposal_step(lat, torch.from_numpy(state), sc, k=K, inp=torch.from_numpy(inp), values=vl, pad=PAD, bos=BOS,
                              eos=EOS, temperature=T, uniforms=torch.from_numpy(u),
                              value_state=torch.from_numpy(vstate) if own else None)
        (r.logq * torch.from_numpy(gq).to(dev) + r.logz * torch.from_numpy(gz).to(dev)).sum().backward()
        sym = r.symbol.cpu().numpy()
        # dense restatement in torch (float64)
        mask = torch.from_numpy(O.mask_out_invalid(em_k, inp, state, 2, 300, PAD, BOS, EOS).astype(np.float64))
        idx = torch.from_numpy(tr_k[np.arange(N), vstate if own else state])  # [N, V] next state per mark
        roff = torch.from_numpy(np.repeat(lat.row_off.astype(np.int64), K))[:, None]
        tsc = torch.from_numpy(scores.astype(np.float64)).requires_grad_(True)
        tvl = torch.from_numpy(values.astype(np.float64)).requires_grad_(True)
        pz = torch.ones(V, dtype=torch.float64); pz[PAD] = 0
        x = ((tsc + tvl[roff + idx]) * pz + mask) / T
        lp = torch.log_softmax(x, dim=1)
        lq = lp[torch.arange(N), torch.from_numpy(sym)]
        lz = torch.logsumexp(x, dim=1)
        (lq * torch.from_numpy(gq.astype(np.float64)) + lz * torch.from_numpy(gz.astype(np.float64))).sum().backward()
        assert np.max(np.abs(r.logq.detach().cpu().numpy() - lq.detach().numpy())) <= 2e-5
        assert np.max(np.abs(sc.grad.cpu().numpy() - tsc.grad.numpy())) <= 2e-5
        assert np.max(np.abs(vl.grad.cpu().numpy() - tvl.grad.numpy())) <= 5e-5


def test_device_prefetcher_overlaps_copies_and_changes_nothing(dev):
    """io.DevicePrefetcher: batches arrive on the device through pinned memory and a side stream;
    results are those of a plain blocking copy."""
    from nfst_amd import ops
    from nfst_amd.lattice import LatticeBatch
    theta = torch.from_numpy(synth.label_scores(3, 64))
    cpu = [LatticeBatch.from_synth([synth.layered_lattice(500 + 7 * i + j, n_states=120 + 30 * j, avg_degree=5.0, vocab=64,
                                                           width=4, span=3) for j in range(4)]) for i in range(5)]
    want = [ops.forward_backward(b.to(dev), theta).logz64.cpu() for b in cpu]
    got = []
    for b in io.DevicePrefetcher(cpu, dev):
        assert b.device.type == "cuda"
        got.append(ops.forward_backward(b, theta).logz64.cpu())
    assert len(got) == len(want) and all(torch.equal(a, b) for a, b in zip(got, want))
    assert list(io.DevicePrefetcher([], dev)) == []
