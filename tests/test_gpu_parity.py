"""Parity of the HIP path (through the C ABI) with the CPU oracle and the golden
fixtures.  Needs a real MI355X: run with ``pytest -m gpu``.

Tolerances: log Z (float64 output) <= 1e-5 absolute everywhere (north_star); the float32 row
outputs log alpha / log beta <= 1e-5 + half a float32 spacing at their magnitude; posteriors
<= 2e-6 absolute at the BASELINE depth (<= 1e-5 for 1500-level lattices); state / arc / label
indices bit-exact.  The largest errors seen are written to gpurun_out/parity_errors.json."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O
from nfst_amd import ops, synth, _lib
from nfst_amd.lattice import LatticeBatch

pytestmark = pytest.mark.gpu
PAD, BOS, EOS = synth.PAD, synth.BOS, synth.EOS
TOL = 1e-5


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


# largest errors seen per check, written to gpurun_out/parity_errors.json at the end of the module
# (DESIGN.md section 5 quotes them)
_ERR = {}


def rec(tag, err):
    err = float(err)
    _ERR[tag] = max(_ERR.get(tag, 0.0), err)
    return err


@pytest.fixture(scope="module", autouse=True)
def _dump_errors():
    yield
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if _ERR and os.path.isdir(out):
        import json
        with open(os.path.join(out, "parity_errors.json"), "w") as f:
            json.dump(dict(sorted(_ERR.items())), f, indent=1)


def oracle_fb(l, theta, arc_scores=None):
    sc = theta[l.label].astype(np.float64)
    if l.weight is not None:
        sc = sc + l.weight.astype(np.float64)
    if arc_scores is not None:
        sc = sc + arc_scores.astype(np.float64)
    return O.forward_backward(l.n_rows, l.src, l.dst, sc), sc


def cmp_rows(got, ref, tol=TOL, tag=None):
    """Row outputs (log alpha, log beta) are float32: beside `tol` they may be off by the rounding
    of the exact value to float32, half a float32 spacing at its magnitude (3e-5 at |log beta| = 600;
    the (mantissa, exponent) pairs `beta_me` carry the full precision).  log Z is checked in float64
    at the plain `tol`."""
    got = np.asarray(got, np.float64)
    inf = np.isneginf(ref)
    assert np.array_equal(np.isneginf(got), inf)
    err = np.abs(got[~inf] - ref[~inf])
    half_ulp = 0.5 * np.spacing(np.abs(ref[~inf]).astype(np.float32)).astype(np.float64)
    if tag:
        rec(tag, err.max())
        rec(tag + "_beyond_f32_rounding", np.maximum(err - half_ulp, 0.0).max())
    assert np.all(err <= tol + half_ulp)


# ----------------------------------------------------------------------------- golden fixtures
@pytest.mark.parametrize("name", ["beta_layered12", "beta_layered40", "beta_layered120", "beta_edit",
                                  "beta_parallel_arc_quirk"])
def test_beta_matches_reference_fixture(dev, golden_dir, name):
    """log beta of the reference's compute_beta_per_sample (scorers.py:692-751)."""
    d = load(golden_dir, name)
    lat = LatticeBatch.from_dense(d["emission"][None], d["transition"][None], device=dev)
    r = ops.backward(lat, torch.from_numpy(d["theta"]))
    got = r.logbeta.cpu().numpy()
    ref = np.log(d["beta_per_sample"].astype(np.float64))
    reach = np.isfinite(got)
    assert reach[0] and reach.sum() == lat.meta_host[0][_lib.META_N_REACH]
    assert np.max(np.abs(got[reach] - ref[reach])) <= TOL
    assert abs(float(r.logz64[0]) - ref[0]) <= TOL and abs(float(r.logz[0]) - ref[0]) <= TOL
    # exp view for legacy consumers (probability domain, like compute_beta returns)
    assert np.allclose(np.exp(got[reach]), d["beta_per_sample"][reach], rtol=2e-5)


@pytest.mark.parametrize("tag", ["pad0", "pad7"])
def test_state_advance_and_masks_bit_exact(dev, golden_dir, tag):
    d = load(golden_dir, "gather")
    K, maxlen = int(d["K"]), int(d["max_length"])
    lat = LatticeBatch.from_dense(d[f"{tag}_emission"], d[f"{tag}_transition"], device=dev)
    for r in range(d[f"{tag}_states"].shape[0]):
        st = torch.from_numpy(d[f"{tag}_states"][r]).to(dev)
        lb = torch.from_numpy(d[f"{tag}_labels"][r]).to(dev)
        nxt = ops.step(lat, st, lb, k=K).cpu().numpy()
        # fixture walkers sit on rows of their own lattice (all reachable); collate padding rows are never visited
        ref = d[f"{tag}_next"][r]
        assert np.array_equal(nxt, ref)
        for L, key in ((5, "mask_len5"), (21, "mask_len21")):
            got = ops.emission_mask(lat, st, k=K, inp=lb, pad=PAD, bos=BOS, eos=EOS, has_to_end=L > maxlen).cpu().numpy()
            assert np.array_equal(got, d[f"{tag}_{key}"][r])


def test_weighted_masks_and_beta_logits_fixture(dev, golden_dir):
    """nfst_emission_mask on float emission tables (the state's row of log weights on top of the legality masks,
    scorers.py:1049-1053) and nfst_beta_logits (scorers.py:584-590) against arrays the reference produced."""
    d = load(golden_dir, "gather")
    K, maxlen = int(d["K"]), int(d["max_length"])
    lat = LatticeBatch.from_dense(d["w_emission"], d["w_transition"], device=dev)
    assert lat.weighted
    # the reference's beta is [B*K, S+1] over the collated row count; the engine's values are per packed row
    S1 = d["w_emission"].shape[1]
    beta = d["w_beta"]
    for r in range(d["w_states"].shape[0]):
        st = torch.from_numpy(d["w_states"][r]).to(dev)
        lb = torch.from_numpy(d["w_labels"][r]).to(dev)
        got = ops.emission_mask(lat, st, k=K, inp=lb, pad=PAD, bos=BOS, eos=EOS, has_to_end=5 > maxlen).cpu().numpy()
        assert np.array_equal(got, d["w_mask_len5"][r])
    # beta logits: one values row per lattice (K = 1 walker per lattice takes the reference's rows 0, K, 2K, ...)
    vals = np.zeros(lat.total_rows, np.float32)
    for b in range(lat.n_lattices):
        r0, n = int(lat.row_off[b]), int(lat.n_rows[b])
        vals[r0:r0 + n] = beta[b * K, :n]
    for r in range(d["w_states"].shape[0]):
        st = torch.from_numpy(d["w_states"][r][::K].copy()).to(dev)
        got = ops.beta_logits(lat, torch.from_numpy(vals).to(dev), st, k=1).cpu().numpy()
        ref = d["w_beta_logits"][r][::K]
        # labels without an arc lead to row 0 in the reference's dense table (transition 0): the engine returns the same row's value
        assert np.array_equal(got, ref), r


def test_iwae_and_wfst_fixture(dev, golden_dir):
    d = load(golden_dir, "iwae")
    B, K, T = d["samples"].shape
    stripped = O.stripping_pad(d["samples"].reshape(B * K, T), PAD)
    log_p = O.wfst_score(d["theta"], stripped, PAD).reshape(B, K)
    lm, log_w = ops.iwae(torch.from_numpy(log_p).to(dev), torch.from_numpy(d["log_q"]).to(dev))
    assert np.max(np.abs(log_w.cpu().numpy() - d["log_w"])) <= TOL
    assert np.max(np.abs(lm.cpu().numpy() - d["log_marginal"])) <= TOL
    # forced walk of the reference's samples: every one is an accepting path, score = sum theta[mark]
    lat = LatticeBatch.from_dense(d["emission"], d["transition"], device=dev)
    marks = np.concatenate([np.full((B, K, 1), BOS, np.int64), d["samples"]], axis=2)
    tot, end = ops.score_paths(lat, torch.from_numpy(d["theta"]), torch.from_numpy(marks))
    assert np.array_equal(end.cpu().numpy(), np.repeat(lat.sink[:, None], K, 1))
    ref = log_p + d["theta"][BOS]
    assert np.max(np.abs(tot.cpu().numpy() - ref)) <= TOL


@pytest.mark.parametrize("tag", ["norm_eval", "norm_eval_temp", "norm_eval_short", "raw_eval", "norm_train_smooth"])
def test_path_logprob_fixture(dev, golden_dir, tag):
    """evaluate_seq_with_temp of the reference (scorers.py:1530-1614), eval and training mode."""
    d = load(golden_dir, "evalseq")
    maxlen, norm, smooth, training, temp = d[tag + "_cfg"]
    got = ops.path_logprob(torch.from_numpy(d["scores"]).to(dev), torch.from_numpy(d["seqs"]).to(dev), pad=PAD, bos=BOS,
                           eos=EOS, max_length=int(maxlen), temp=float(temp), normalize=bool(norm),
                           smoothing=float(smooth) if training else 0.0).cpu().numpy()
    ref = d[tag]
    same_special = (np.isnan(ref) & np.isnan(got)) | (np.isinf(ref) & (ref == got))
    fin = np.isfinite(ref)
    assert np.all(same_special | fin)
    assert np.max(np.abs(got[fin] - ref[fin]) / np.maximum(1.0, np.abs(ref[fin]))) <= 2e-5


# ----------------------------------------------------------------------------- oracle parity
def _mixed_batch():
    return [
        synth.layered_lattice(3, n_states=30, avg_degree=3.0, vocab=64, width=4, span=2),
        synth.layered_lattice(4, n_states=300, avg_degree=8.0, vocab=64, width=9, span=5),
        synth.layered_lattice(5, n_states=90, avg_degree=5.0, vocab=64, width=1, span=6),
        synth.edit_lattice([10, 11, 12, 13, 14], [20, 21, 22, 23], vocab=64, seed=2),
        synth.layered_lattice(6, n_states=700, avg_degree=10.0, vocab=64, width=16, span=8),
        synth._finish(2, 64, [0], [EOS], [1]),
    ]


@pytest.mark.parametrize("opts", [dict(), dict(slots_per_lane=1), dict(slots_per_lane=2), dict(slots_per_lane=4),
                                  dict(slots_per_lane=4, no_compact=True)])
def test_forward_backward_matches_oracle(dev, opts):
    lats = _mixed_batch()
    theta = synth.label_scores(7, 64)
    lat = LatticeBatch.from_synth(lats, device=dev, **opts)
    r = ops.forward_backward(lat, torch.from_numpy(theta), want_grad_theta=True)
    la, lb, post = r.logalpha.cpu().numpy(), r.logbeta.cpu().numpy(), r.posterior.cpu().numpy()
    gth = r.grad_theta.cpu().numpy()
    for b, l in enumerate(lats):
        o, _ = oracle_fb(l, theta)
        r0, a0 = int(lat.row_off[b]), int(lat.arc_off[b])
        cmp_rows(la[r0:r0 + l.n_rows], o["logalpha"])
        cmp_rows(lb[r0:r0 + l.n_rows], o["logbeta"])
        assert abs(float(r.logz64[b]) - o["logZ"]) <= TOL
        assert abs(float(r.logz[b]) - o["logZ"]) <= TOL
        assert np.max(np.abs(post[a0:a0 + l.n_arcs] - o["posterior"])) <= 2e-6
        ref_g = np.bincount(l.label, weights=o["posterior"], minlength=64)
        assert np.max(np.abs(gth[b] - ref_g)) <= 1e-4
    # beta-only kernel agrees with the fused one bit for bit
    rb = ops.backward(lat, torch.from_numpy(theta))
    assert torch.equal(rb.logbeta, r.logbeta) and torch.equal(rb.logz64, r.logz64)


def test_weighted_tables_and_arc_scores(dev):
    """float emission tables (get_state_mask_pynini weighted=True, scorers.py:1011-1027)
    and caller-supplied per-arc scores; per-lattice theta [B, V]."""
    lats = [synth.layered_lattice(s, n_states=150 + 20 * s, avg_degree=6.0, vocab=48, width=7, span=3, weighted=True)
            for s in range(4)]
    em, tr = synth.collate_dense([l.dense(weighted=True) for l in lats])
    lat = LatticeBatch.from_dense(em, tr, device=dev)
    assert lat.weighted == 1
    rng = np.random.default_rng(0)
    theta = rng.normal(-2.0, 0.7, size=(len(lats), 48)).astype(np.float32)
    arc_scores = rng.normal(0.0, 0.3, size=lat.total_arcs).astype(np.float32)
    r = ops.forward_backward(lat, torch.from_numpy(theta), arc_scores=torch.from_numpy(arc_scores))
    g = ops.gather_label_scores(lat, torch.from_numpy(theta), torch.from_numpy(arc_scores)).cpu().numpy()
    for b, l in enumerate(lats):
        a0 = int(lat.arc_off[b])
        asc = arc_scores[a0:a0 + l.n_arcs]
        o, sc = oracle_fb(l, theta[b], asc)
        assert np.max(np.abs(g[a0:a0 + l.n_arcs] - sc)) <= 2e-6
        r0 = int(lat.row_off[b])
        cmp_rows(r.logbeta.cpu().numpy()[r0:r0 + l.n_rows], o["logbeta"])
        cmp_rows(r.logalpha.cpu().numpy()[r0:r0 + l.n_rows], o["logalpha"])
        assert abs(float(r.logz64[b]) - o["logZ"]) <= TOL
        assert np.max(np.abs(r.posterior.cpu().numpy()[a0:a0 + l.n_arcs] - o["posterior"])) <= 2e-6


def test_arc_scores_taken_as_a_slice_of_a_larger_tensor(dev):
    """``arc_scores`` that is a view into a larger score tensor (contiguous, but its first element is not
    16-byte aligned) gives the bits of an aligned copy: every op that takes per-arc scores."""
    lats = [synth.layered_lattice(60 + i, n_states=150, avg_degree=5.0, vocab=48, width=6, span=3, weighted=(i % 2 == 0)) for i in range(2)]
    for group in ([lats[0]], [lats[1]]):
        lat = LatticeBatch.from_synth(group, device=dev)
        th = torch.from_numpy(synth.label_scores(3, 48))
        big = torch.randn(lat.total_arcs + 11, device=dev) * 0.3
        for off in (1, 3, 5):
            view = big[off:off + lat.total_arcs]
            assert view.data_ptr() % 16 != 0
            ref = view.clone()
            a, b = ops.forward_backward(lat, th, arc_scores=view), ops.forward_backward(lat, th, arc_scores=ref)
            assert torch.equal(a.logz64, b.logz64) and torch.equal(a.posterior, b.posterior) and torch.equal(a.logalpha, b.logalpha)
            assert torch.equal(ops.backward(lat, th, arc_scores=view).logz64, ops.backward(lat, th, arc_scores=ref).logz64)
            va, vb = ops.viterbi(lat, th, arc_scores=view), ops.viterbi(lat, th, arc_scores=ref)
            assert torch.equal(va.best, vb.best) and torch.equal(va.paths, vb.paths)
            sa, sb = ops.sample_paths(lat, th, 4, arc_scores=view, seed=5), ops.sample_paths(lat, th, 4, arc_scores=ref, seed=5)
            assert torch.equal(sa.paths, sb.paths) and torch.equal(sa.logq, sb.logq)
            assert torch.equal(ops.gather_label_scores(lat, th, arc_scores=view), ops.gather_label_scores(lat, th, arc_scores=ref))
            z = ops.log_z(lat, th, view.clone().requires_grad_(True))
            assert torch.equal(z.detach(), a.logz)


def test_extreme_scores_do_not_overflow(dev):
    """The reference's probability-domain beta overflows float32 once log Z > 88
    (SURVEY.md section 6); the (mantissa, exponent) semiring must not."""
    l = synth.layered_lattice(11, n_states=400, avg_degree=8.0, vocab=64, width=8, span=4)
    for mean in (6.0, -40.0):
        theta = synth.label_scores(3, 64, mean=mean, std=1.0)
        lat = LatticeBatch.from_synth([l], device=dev)
        r = ops.forward_backward(lat, torch.from_numpy(theta))
        o, _ = oracle_fb(l, theta)
        assert abs(o["logZ"]) > 200
        assert rec("extreme_scores_logz", abs(float(r.logz64[0]) - o["logZ"])) <= TOL  # |log Z| > 200, absolute
        assert rec("extreme_scores_post", np.max(np.abs(r.posterior.cpu().numpy() - o["posterior"]))) <= 5e-6
    # a forbidden label (-inf) removes its arcs
    theta = synth.label_scores(3, 64)
    theta[l.label[5]] = -np.inf
    r = ops.forward_backward(LatticeBatch.from_synth([l], device=dev), torch.from_numpy(theta))
    o, _ = oracle_fb(l, theta)
    assert abs(float(r.logz64[0]) - o["logZ"]) <= TOL


def test_huge_degree_and_deep_lattices(dev):
    V = 256
    src = [0] + [1] * 200 + list(range(2, 202)) + [202]
    lab = [BOS] + list(range(3, 203)) + [5] * 200 + [EOS]
    dst = [1] + list(range(2, 202)) + [202] * 200 + [203]
    star = synth._finish(204, V, src, lab, dst)
    deep = synth.layered_lattice(5, n_states=1500, avg_degree=10.0, vocab=V, width=1, span=8)
    theta = synth.label_scores(2, V)
    for opts in (dict(), dict(slots_per_lane=1), dict(slots_per_lane=2), dict(slots_per_lane=4), dict(group_mode=1),
                 dict(group_mode=1, slots_per_lane=1), dict(group_mode=2)):
        lat = LatticeBatch.from_synth([star, deep], device=dev, **opts)
        r = ops.forward_backward(lat, torch.from_numpy(theta))
        for b, l in enumerate([star, deep]):
            o, _ = oracle_fb(l, theta)
            assert abs(float(r.logz64[b]) - o["logZ"]) <= TOL
            a0 = int(lat.arc_off[b])
            assert np.max(np.abs(r.posterior.cpu().numpy()[a0:a0 + l.n_arcs] - o["posterior"])) <= 2e-6


def test_autograd_gives_posteriors(dev):
    lats = _mixed_batch()[:5]
    lat = LatticeBatch.from_synth(lats, device=dev)
    theta = torch.from_numpy(synth.label_scores(7, 64)).to(dev).requires_grad_(True)
    arc_scores = torch.zeros(lat.total_arcs, device=dev, requires_grad=True)
    z = ops.log_z(lat, theta, arc_scores)
    wts = torch.arange(1, len(lats) + 1, device=dev, dtype=torch.float32)
    (z * wts).sum().backward()
    ref_theta = np.zeros(64)
    for b, l in enumerate(lats):
        o, _ = oracle_fb(l, theta.detach().cpu().numpy())
        a0 = int(lat.arc_off[b])
        assert np.max(np.abs(arc_scores.grad.cpu().numpy()[a0:a0 + l.n_arcs] - (b + 1) * o["posterior"])) <= 1e-5
        ref_theta += (b + 1) * np.bincount(l.label, weights=o["posterior"], minlength=64)
    assert np.max(np.abs(theta.grad.cpu().numpy() - ref_theta)) <= 1e-3


def test_viterbi_bit_exact(dev):
    lats = _mixed_batch()
    theta = synth.label_scores(8, 64)
    lat = LatticeBatch.from_synth(lats, device=dev)
    r = ops.viterbi(lat, torch.from_numpy(theta), pad=PAD)
    for b, l in enumerate(lats):
        best, path, arcs = O.viterbi(l.n_rows, l.src, l.label, l.dst, theta[l.label], 4000)
        n = int(r.lengths[b])
        assert n == len(path)
        assert np.float32(best) == r.best.cpu().numpy()[b]  # same float32 adds, same order
        assert np.array_equal(r.paths.cpu().numpy()[b, :n], path)
        assert np.array_equal(r.arcs.cpu().numpy()[b, :n] - int(lat.arc_off[b]), arcs)
        assert np.all(r.paths.cpu().numpy()[b, n:] == PAD)
    # a state with 200 out-arcs: continuation pieces (wide groups) and partial groups + combine piece
    # (narrow groups) must hand the best arc through their unit-label records
    V = 256
    src = [0] + [1] * 200 + list(range(2, 202)) + [202]
    lab = [BOS] + list(range(3, 203)) + [5] * 200 + [EOS]
    dst = [1] + list(range(2, 202)) + [202] * 200 + [203]
    star = synth._finish(204, V, src, lab, dst)
    theta = synth.label_scores(4, V)
    best, path, arcs = O.viterbi(star.n_rows, star.src, star.label, star.dst, theta[star.label], 4000)
    for opts in (dict(group_mode=1), dict(group_mode=2), dict(group_mode=1, slots_per_lane=1), dict(group_mode=2, slots_per_lane=1),
                 dict(group_mode=1, slots_per_lane=4, no_compact=True)):
        lat = LatticeBatch.from_synth([star], device=dev, **opts)
        r = ops.viterbi(lat, torch.from_numpy(theta), pad=PAD)
        n = int(r.lengths[0])
        assert n == len(path) and np.float32(best) == r.best.cpu().numpy()[0]
        assert np.array_equal(r.paths.cpu().numpy()[0, :n], path)
        assert np.array_equal(r.arcs.cpu().numpy()[0, :n], arcs)


def test_posterior_sampling(dev):
    lats = _mixed_batch()[:5]
    theta = synth.label_scores(9, 64)
    lat = LatticeBatch.from_synth(lats, device=dev)
    K, T = 512, int(lat.depth.max()) + 1
    rng = np.random.default_rng(1)
    u = rng.random((len(lats), K, T)).astype(np.float32)
    s = ops.sample_paths(lat, torch.from_numpy(theta), K, max_len=T, uniforms=torch.from_numpy(u), pad=PAD)
    paths, arcs, lens, logq = (x.cpu().numpy() for x in (s.paths, s.arcs, s.lengths, s.logq))
    for b, l in enumerate(lats):
        o, sc = oracle_fb(l, theta)
        ref = O.sample_paths(l.n_rows, l.src, l.label, l.dst, sc, o["logbeta"], u[b].astype(np.float64), PAD)
        safe = ref["margin"] > 1e-5  # walks whose uniforms stay clear of every CDF boundary
        assert safe.mean() > 0.97
        assert np.array_equal(paths[b][safe], ref["paths"][safe])
        assert np.array_equal(lens[b][safe], ref["lengths"][safe])
        assert np.array_equal((arcs[b] - np.where(arcs[b] >= 0, int(lat.arc_off[b]), 0))[safe], ref["arcs"][safe])
        # every walk (safe or not) is an accepting path with log q = score - log Z
        for k in range(0, K, 37):
            a = arcs[b, k, :lens[b, k]] - int(lat.arc_off[b])
            assert l.src[a[0]] == 0 and l.dst[a[-1]] == l.n_rows - 1 and np.all(l.dst[a[:-1]] == l.src[a[1:]])
            assert abs(sc[a].sum() - o["logZ"] - logq[b, k]) <= 2e-5
    # IWAE with the exact posterior as proposal has zero variance: every log w == log Z
    tot, end = ops.score_paths(lat, torch.from_numpy(theta), s.paths)
    lm, log_w = ops.iwae(tot, s.logq)
    z = s.logz.cpu().numpy()
    assert np.max(np.abs(log_w.cpu().numpy() - z[:, None])) <= 5e-5
    assert np.max(np.abs(lm.cpu().numpy() - z)) <= 5e-5
    # the Philox path is deterministic in the seed and follows the posterior
    a = ops.sample_paths(lat, torch.from_numpy(theta), 2048, seed=11, pad=PAD)
    b2 = ops.sample_paths(lat, torch.from_numpy(theta), 2048, seed=11, pad=PAD)
    c = ops.sample_paths(lat, torch.from_numpy(theta), 2048, seed=12, pad=PAD)
    assert torch.equal(a.paths, b2.paths) and not torch.equal(a.paths, c.paths)
    for b, l in enumerate(lats[:2]):
        o, _ = oracle_fb(l, theta)
        ar = a.arcs.cpu().numpy()[b]
        cnt = np.bincount(ar[ar >= 0] - int(lat.arc_off[b]), minlength=l.n_arcs) / 2048.0
        assert np.max(np.abs(cnt - o["posterior"])) < 0.06


def test_posterior_sampling_kernel_modes(dev):
    """The three ways k_sample reads a lattice -- cumulative arc probabilities precomputed in LDS (path_arcs wanted,
    CSR fits), CSR staged in LDS (no path_arcs), global memory (a lattice too large for LDS) -- draw the same paths
    from the same uniforms wherever u stays clear of a CDF boundary."""
    theta = synth.label_scores(9, 64)
    lats = _mixed_batch()[:4]
    lat = LatticeBatch.from_synth(lats, device=dev)
    K, T = 96, int(lat.depth.max()) + 1
    u = torch.from_numpy(np.random.default_rng(3).random((len(lats), K, T)).astype(np.float32))
    a = ops.sample_paths(lat, torch.from_numpy(theta), K, max_len=T, uniforms=u, pad=PAD)
    b = ops.sample_paths(lat, torch.from_numpy(theta), K, max_len=T, uniforms=u, pad=PAD, want_arcs=False)
    assert b.arcs is None
    same = (a.paths == b.paths).all(dim=2)
    assert same.float().mean() > 0.995  # (running sums against a DPP scan: the last bit of a CDF value may differ)
    assert torch.equal(a.lengths[same], b.lengths[same])
    assert float((a.logq[same] - b.logq[same]).abs().max()) <= 2e-5
    # 3000 states x ~10 arcs: 6 bytes per arc + the beta rows do not fit 160 KiB
    big = synth.layered_lattice(41, n_states=3000, avg_degree=10.0, vocab=64, width=16, span=6)
    assert big.n_arcs * 6 + big.n_rows * 12 > 160 * 1024
    lat = LatticeBatch.from_synth([big], device=dev)
    K, T = 24, int(lat.depth.max()) + 1
    u = np.random.default_rng(4).random((1, K, T)).astype(np.float32)
    for want in (True, False):
        s = ops.sample_paths(lat, torch.from_numpy(theta), K, max_len=T, uniforms=torch.from_numpy(u), pad=PAD, want_arcs=want)
        o, sc = oracle_fb(big, theta)
        ref = O.sample_paths(big.n_rows, big.src, big.label, big.dst, sc, o["logbeta"], u[0].astype(np.float64), PAD)
        safe = ref["margin"] > 1e-5
        assert safe.mean() > 0.9
        assert np.array_equal(s.paths.cpu().numpy()[0][safe], ref["paths"][safe])
        assert np.array_equal(s.lengths.cpu().numpy()[0][safe], ref["lengths"][safe])
        if want:
            assert np.array_equal(s.arcs.cpu().numpy()[0][safe], ref["arcs"][safe])  # (one lattice: arc_off = 0)


@pytest.mark.parametrize("want_arcs", [True, False])
def test_posterior_sampling_with_table_weights_and_arc_scores(dev, want_arcs):
    """Weighted tables and caller-supplied per-arc scores enter the arc probabilities (and log q) of the walks."""
    lats = [synth.layered_lattice(81, n_states=150, avg_degree=6.0, vocab=64, width=6, span=3, weighted=True),
            synth.layered_lattice(82, n_states=400, avg_degree=10.0, vocab=64, width=12, span=5, weighted=True)]
    theta = synth.label_scores(13, 64)
    lat = LatticeBatch.from_synth(lats, device=dev)
    rng = np.random.default_rng(8)
    asc = (0.5 * rng.standard_normal(lat.total_arcs)).astype(np.float32)
    K, T = 64, int(lat.depth.max()) + 1
    u = rng.random((len(lats), K, T)).astype(np.float32)
    s = ops.sample_paths(lat, torch.from_numpy(theta), K, arc_scores=torch.from_numpy(asc), max_len=T,
                         uniforms=torch.from_numpy(u), pad=PAD, want_arcs=want_arcs)
    for b, l in enumerate(lats):
        a0 = int(lat.arc_off[b])
        o, sc = oracle_fb(l, theta, asc[a0:a0 + l.n_arcs])
        ref = O.sample_paths(l.n_rows, l.src, l.label, l.dst, sc, o["logbeta"], u[b].astype(np.float64), PAD)
        safe = ref["margin"] > 1e-5
        assert safe.mean() > 0.95
        assert np.array_equal(s.paths.cpu().numpy()[b][safe], ref["paths"][safe])
        assert np.array_equal(s.lengths.cpu().numpy()[b][safe], ref["lengths"][safe])
        if want_arcs:
            arcs = s.arcs.cpu().numpy()[b]
            assert np.array_equal((arcs - np.where(arcs >= 0, a0, 0))[safe], ref["arcs"][safe])
            for k in np.flatnonzero(safe)[::9]:
                a = arcs[k, :s.lengths.cpu().numpy()[b, k]] - a0
                assert abs(sc[a].sum() - o["logZ"] - float(s.logq[b, k])) <= 3e-5


def test_beta_logits_gather(dev):
    lats = _mixed_batch()[:3]
    K = 4
    em, tr = synth.collate_dense([l.dense() for l in lats])
    lat = LatticeBatch.from_dense(em, tr, device=dev)
    theta = synth.label_scores(3, 64)
    r = ops.backward(lat, torch.from_numpy(theta))
    rng = np.random.default_rng(5)
    # walkers sit on reachable states
    states = np.array([rng.choice(np.unique(lats[n // K].src)) for n in range(len(lats) * K)], np.int64)
    got = ops.beta_logits(lat, r.logbeta, torch.from_numpy(states), k=K).cpu().numpy()
    beta = lat.rows_view(r.logbeta).cpu().numpy()
    ref = O.beta_logits(O.expand_k(tr, K), np.repeat(beta, K, 0), states)
    assert np.array_equal(got, ref)


# ----------------------------------------------------------------------------- BASELINE size
def test_baseline_batch_properties_and_oracle(dev):
    """256 synthetic lattices of ~2k states / ~20k arcs (BASELINE.json configs[1])."""
    B = 256
    lats = synth.bench_batch(B)
    theta = synth.label_scores(1, 256)
    lat = LatticeBatch.from_synth(lats, device=dev)
    assert 4.5e6 < lat.total_arcs < 5.6e6
    r = ops.forward_backward(lat, torch.from_numpy(theta))
    n_rows, arc_off, src, label, dst, w = synth.batch_arcs(lats)
    zref, pref = O.forward_backward_batch(n_rows, arc_off, src, label, dst, None, theta, n_threads=8)
    assert rec("baseline_logz64", np.max(np.abs(r.logz64.cpu().numpy() - zref))) <= TOL
    assert rec("baseline_logz32", np.max(np.abs(r.logz.cpu().numpy() - zref))) <= TOL
    post = r.posterior.cpu().numpy().astype(np.float64)
    assert rec("baseline_post", np.max(np.abs(post - pref))) <= 2e-6
    la = r.logalpha.cpu().numpy()
    # size-independent identities: Z from alpha == Z from beta; unit flow out of the start and
    # into the sink; flow conservation at every inner state
    sink_rows = lat.row_off + lat.sink
    assert np.max(np.abs(la[sink_rows] - r.logz.cpu().numpy())) <= TOL
    lat_of = np.repeat(np.arange(B), np.diff(arc_off))
    gsrc = src + lat.row_off[lat_of]
    gdst = dst + lat.row_off[lat_of]
    nl = src != dst
    outflow = np.bincount(gsrc[nl], weights=post[nl], minlength=lat.total_rows)
    inflow = np.bincount(gdst[nl], weights=post[nl], minlength=lat.total_rows)
    assert np.max(np.abs(outflow[lat.row_off] - 1.0)) <= 2e-5
    assert np.max(np.abs(inflow[sink_rows] - 1.0)) <= 2e-5
    inner = np.ones(lat.total_rows, bool); inner[lat.row_off] = False; inner[sink_rows] = False
    assert np.max(np.abs(inflow[inner] - outflow[inner])) <= 2e-5
    # run-to-run determinism of everything but the LDS label histogram
    r2 = ops.forward_backward(lat, torch.from_numpy(theta))
    assert torch.equal(r.posterior, r2.posterior) and torch.equal(r.logz64, r2.logz64)


def test_configs3_shard_1024_lattices(dev):
    """BASELINE.json configs[3]: 8192 lattices over 8 GPUs = one shard of 1024 x (~2k states /
    ~20k arcs) per GPU.  More lattices than 2 x CUs: the 256-thread flavour (two workgroups per CU,
    self-loading decoders).  Oracle (float64) on every 16th lattice at the strict 1e-5 / 2e-6; the
    size-independent flow identities on all of them; and the first 256 / 512 lattices must come out
    with the bits of the 1024-thread (one lattice per CU) and 512-thread flavours."""
    B = 1024
    lats = synth.bench_batch(B)
    theta_np = synth.label_scores(1, 256)
    theta = torch.from_numpy(theta_np).to(dev)
    quarters = [LatticeBatch.from_synth(lats[i:i + 256]) for i in range(0, B, 256)]  # packed once per quarter ...
    lat = LatticeBatch.concat(quarters, device=dev)  # ... the shard by concatenation (bit-identical to packing it whole)
    assert lat.n_lattices == B and 18e6 < lat.total_arcs < 23e6
    r = ops.forward_backward(lat, theta)
    z64, z32 = r.logz64.cpu().numpy(), r.logz.cpu().numpy()
    post = r.posterior.cpu().numpy()
    sample = list(range(0, B, 16))
    n_rows, arc_off, src, label, dst, w = synth.batch_arcs([lats[i] for i in sample])
    zref, pref = O.forward_backward_batch(n_rows, arc_off, src, label, dst, None, theta_np, n_threads=8)
    assert rec("configs3_logz64", np.max(np.abs(z64[sample] - zref))) <= TOL
    assert rec("configs3_logz32", np.max(np.abs(z32[sample] - zref))) <= TOL
    for j, b in enumerate(sample):
        a0 = int(lat.arc_off[b])
        got = post[a0:a0 + lats[b].n_arcs].astype(np.float64)
        assert rec("configs3_post", np.max(np.abs(got - pref[arc_off[j]:arc_off[j + 1]]))) <= 2e-6
    # identities on all 1024: Z from alpha == Z from beta; unit flow out of the start and into the
    # sink; flow conservation at every inner state
    la = r.logalpha.cpu().numpy()
    sink_rows = lat.row_off + lat.sink
    assert rec("configs3_z_alpha_vs_beta", np.max(np.abs(la[sink_rows] - z32))) <= TOL
    gsrc = lat.arc_src.cpu().numpy().astype(np.int64)
    gdst = lat.arc_dst.cpu().numpy().astype(np.int64)
    off = np.repeat(lat.row_off.astype(np.int64), lat.n_arcs)
    nl = gsrc != gdst
    p64 = post.astype(np.float64)
    outflow = np.bincount((gsrc + off)[nl], weights=p64[nl], minlength=lat.total_rows)
    inflow = np.bincount((gdst + off)[nl], weights=p64[nl], minlength=lat.total_rows)
    assert np.max(np.abs(outflow[lat.row_off] - 1.0)) <= 2e-5
    assert np.max(np.abs(inflow[sink_rows] - 1.0)) <= 2e-5
    inner = np.ones(lat.total_rows, bool); inner[lat.row_off] = False; inner[sink_rows] = False
    assert rec("configs3_flow_conservation", np.max(np.abs(inflow[inner] - outflow[inner]))) <= 2e-5
    # flavours: 256 lattices -> 1024 threads, 512 -> 512 threads, 1024 -> 256 threads: same bits
    for n in (256, 512):
        part = LatticeBatch.concat(quarters[: n // 256], device=dev)
        rp = ops.forward_backward(part, theta)
        A, R = part.total_arcs, part.total_rows
        assert torch.equal(rp.logz64, r.logz64[:n]) and torch.equal(rp.posterior, r.posterior[:A])
        assert torch.equal(rp.logalpha, r.logalpha[:R]) and torch.equal(rp.logbeta, r.logbeta[:R])
    # the fused loss of the shard
    total = torch.zeros(3, dtype=torch.float64, device=dev)
    ops.forward_backward(lat, theta, out=r, total=total, total_slot=0)
    assert abs(float(total[0]) - float(r.logz64.sum())) <= 1e-9 * abs(float(r.logz64.sum()))


@pytest.mark.parametrize("V,T", [(300, 37), (1000, 11), (301, 9), (64, 300), (100, 33), (128, 16), (256, 19), (384, 40),
                                 (500, 10), (600, 21), (640, 7), (644, 9), (36, 9), (180, 23), (332, 18), (440, 13),
                                 (700, 9), (900, 5), (1024, 6), (1028, 4)])
def test_path_logprob_matches_oracle_all_variants(dev, V, T):
    """16-byte streaming variants (V % 4 == 0, V <= 1024: one row per wave, half wave or quarter wave, 1 to 7
    16-byte slots per lane; row counts that leave the last slot half empty) and the scalar fallback."""
    rng = np.random.default_rng(V)
    N = 5
    seqs = np.full((N, T), PAD, dtype=np.int64)
    for n in range(N):
        L = int(rng.integers(2, T - 1))
        seqs[n, :L] = rng.integers(3, V, size=L)
        seqs[n, L] = EOS
    scores = rng.normal(0, 2.0, size=(N, T, V)).astype(np.float32)
    for norm, temp, maxlen, smooth in ((True, 1.0, 1000, 0.0), (True, 0.6, 1000, 0.0), (False, 1.0, 1000, 0.0),
                                       (True, 1.0, T // 2, 0.0), (True, 1.0, 1000, 0.1), (False, 0.8, 1000, 0.2)):
        ref = O.evaluate_seq(scores, seqs, PAD, BOS, EOS, maxlen, temp=temp, normalize=norm, training=smooth > 0,
                             smoothing=smooth)
        got = ops.path_logprob(torch.from_numpy(scores).to(dev), torch.from_numpy(seqs).to(dev), pad=PAD, bos=BOS,
                               eos=EOS, max_length=maxlen, temp=temp, normalize=norm,
                               smoothing=smooth).cpu().numpy().astype(np.float64)
        fin = np.isfinite(ref)
        assert np.array_equal(np.isnan(ref), np.isnan(got)) and np.array_equal(np.isinf(ref), np.isinf(got))
        assert np.max(np.abs(got[fin] - ref[fin]) / np.maximum(1.0, np.abs(ref[fin]))) <= 2e-5

GRAD_TAGS = ["norm_eval", "norm_eval_temp", "norm_eval_short", "raw_eval", "norm_train_smooth", "norm_train_smooth_temp",
             "raw_train_smooth"]


@pytest.mark.parametrize("V", [20, 22])
@pytest.mark.parametrize("tag", GRAD_TAGS)
def test_path_logprob_backward_fixture(dev, golden_dir, V, tag):
    """Value and gradient of evaluate_seq_with_temp as torch.autograd gives them through the reference's
    own code (scorers.py:1564-1611; lightning.py:511-516 trains p~ through it).  V = 20: the 16-byte
    streaming kernels, V = 22: the scalar fallback."""
    d = load(golden_dir, "evalseq_grad")
    maxlen, norm, smooth, training, temp = d[f"v{V}_{tag}_cfg"]
    sc = torch.from_numpy(d[f"v{V}_scores"]).to(dev).requires_grad_(True)
    val = ops.path_logprob(sc, torch.from_numpy(d[f"v{V}_seqs"]).to(dev), pad=PAD, bos=BOS, eos=EOS, max_length=int(maxlen),
                           temp=float(temp), normalize=bool(norm), smoothing=float(smooth) if training else 0.0)
    (val * torch.from_numpy(d[f"v{V}_g"]).to(dev)).sum().backward()
    ref_v, ref_g = d[f"v{V}_{tag}"], d[f"v{V}_{tag}_grad"]
    assert np.max(np.abs(val.detach().cpu().numpy() - ref_v) / np.maximum(1.0, np.abs(ref_v))) <= 2e-5
    assert rec("path_logprob_grad_fixture", np.max(np.abs(sc.grad.cpu().numpy() - ref_g) / np.maximum(1.0, np.abs(ref_g)))) <= 2e-5


@pytest.mark.parametrize("V", [20, 22, 300])
def test_gpt2_logprob_fixture(dev, golden_dir, V):
    """GPT2Wrapper.forward on supplied logits (transformer.py:38-52): pad logit -1e8, no legality masks,
    gold shifted with a trailing pad; value and gradient produced by the reference."""
    d = load(golden_dir, "gpt2")
    lg = torch.from_numpy(d[f"v{V}_logits"]).to(dev).requires_grad_(True)
    val = ops.gpt2_logprob(lg, torch.from_numpy(d[f"v{V}_x"]).to(dev), pad=PAD)
    (val * torch.from_numpy(d[f"v{V}_g"]).to(dev)).sum().backward()
    ref_v, ref_g = d[f"v{V}_value"], d[f"v{V}_grad"]
    assert np.max(np.abs(val.detach().cpu().numpy() - ref_v) / np.maximum(1.0, np.abs(ref_v))) <= 2e-5
    assert np.max(np.abs(lg.grad.cpu().numpy() - ref_g)) <= 2e-5


@pytest.mark.parametrize("V,T", [(300, 37), (1000, 11), (301, 9), (64, 40), (100, 33), (128, 16), (256, 19), (384, 40),
                                 (500, 10), (600, 21), (36, 9), (180, 23), (440, 13), (700, 9), (900, 5), (1024, 6), (1028, 4)])
def test_path_logprob_backward_matches_oracle_all_variants(dev, V, T):
    """Every kernel variant of the backward pass against the float64 restatement (pinned by the
    fixtures above): evaluation, temperature, short max_length, raw scores, label smoothing; GPT2 mode."""
    rng = np.random.default_rng(1000 + V)
    N = 5
    seqs = np.full((N, T), PAD, dtype=np.int64)
    for n in range(N):
        L = int(rng.integers(2, T - 1))
        seqs[n, :L] = rng.integers(3, V, size=L)
        seqs[n, L] = EOS
    scores = rng.normal(0, 2.0, size=(N, T, V)).astype(np.float32)
    g = rng.normal(0, 1.0, size=N).astype(np.float32)
    tsc, tseq, tg = torch.from_numpy(scores).to(dev), torch.from_numpy(seqs).to(dev), torch.from_numpy(g).to(dev)
    for norm, temp, maxlen, smooth in ((True, 1.0, 1000, 0.0), (True, 0.6, 1000, 0.0), (False, 1.0, 1000, 0.0),
                                       (True, 1.0, T, 0.0), (True, 1.0, 1000, 0.1), (False, 0.8, 1000, 0.2)):
        ref = O.evaluate_seq_grad(scores, seqs, g, PAD, BOS, EOS, maxlen, temp=temp, normalize=norm, training=smooth > 0,
                                  smoothing=smooth)
        sc = tsc.clone().requires_grad_(True)
        val = ops.path_logprob(sc, tseq, pad=PAD, bos=BOS, eos=EOS, max_length=maxlen, temp=temp, normalize=norm, smoothing=smooth)
        (val * tg).sum().backward()
        got = sc.grad.cpu().numpy().astype(np.float64)
        assert np.all(np.isfinite(got))
        assert rec("path_logprob_grad_oracle", np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref)))) <= 2e-5
    x = seqs[:, :-1]
    refv, refg = O.gpt2_logprob(scores, x, PAD, g)
    lg = tsc.clone().requires_grad_(True)
    val = ops.gpt2_logprob(lg, torch.from_numpy(x).to(dev), pad=PAD)
    (val * tg).sum().backward()
    assert np.max(np.abs(val.detach().cpu().numpy() - refv) / np.maximum(1.0, np.abs(refv))) <= 2e-5
    assert np.max(np.abs(lg.grad.cpu().numpy() - refg)) <= 2e-5


def test_large_lattice_falls_back_to_small_rings(dev):
    """A lattice close to the LDS limit (alpha + beta = 16 B per state of the CU's 160 KiB) leaves
    room for the smallest rings only (3 decoded + 5 staging slots, self-loading decoder)."""
    V = 256
    big = synth.layered_lattice(3, n_states=7800, avg_degree=6.0, vocab=V, width=32, span=4)
    small = synth.layered_lattice(4, n_states=300, avg_degree=6.0, vocab=V, width=8, span=4)
    theta = synth.label_scores(5, V)
    lat = LatticeBatch.from_synth([big, small], device=dev)
    assert lat.lds_bytes() <= 160 * 1024
    r = ops.forward_backward(lat, torch.from_numpy(theta))
    rb = ops.backward(lat, torch.from_numpy(theta))
    for b, l in enumerate([big, small]):
        o, _ = oracle_fb(l, theta)
        r0, a0 = int(lat.row_off[b]), int(lat.arc_off[b])
        assert rec("large_lattice_logz", abs(float(r.logz64[b]) - o["logZ"])) <= TOL
        assert abs(float(rb.logz64[b]) - o["logZ"]) <= TOL
        cmp_rows(r.logbeta.cpu().numpy()[r0:r0 + l.n_rows], o["logbeta"], tag="large_lattice_logbeta")
        assert rec("large_lattice_post", np.max(np.abs(r.posterior.cpu().numpy()[a0:a0 + l.n_arcs] - o["posterior"]))) <= 5e-6
    # beyond the limit the launch is refused with an error code, not a fault (the three-wave pipeline: alpha +
    # beta + the smallest rings of an 8180-state lattice exceed a CU's LDS)
    huge = synth.layered_lattice(6, n_states=8180, avg_degree=4.0, vocab=V, width=32, span=4)
    lat = LatticeBatch.from_synth([huge], device=dev)
    with pytest.raises(_lib.NfstError) as e:
        ops.forward_backward(lat, torch.from_numpy(theta))
    assert e.value.code == -6  # NFST_ERR_LIMIT
    # the fused sweeps (more lattices than CUs, compact programs) keep no rings: the same lattice fits
    many = LatticeBatch.concat([LatticeBatch.from_synth([huge], slots_per_lane=4)] + [LatticeBatch.from_synth([small], slots_per_lane=4)] * 300,
                               device=dev)
    assert many.reserved0 & 1  # every program compact
    r = ops.forward_backward(many, torch.from_numpy(theta))
    o, _ = oracle_fb(huge, theta)
    assert rec("huge_lattice_fused_logz", abs(float(r.logz64[0]) - o["logZ"])) <= TOL
    o, _ = oracle_fb(small, theta)
    assert np.max(np.abs(r.logz64.cpu().numpy()[1:] - o["logZ"])) <= TOL


def test_huge_negative_scores_saturate_instead_of_wrapping(dev):
    """Scores like -1e9 (a common "mask" value) must behave as zero weights: their exponents,
    added up along a path, would leave the int32 range."""
    l = synth.layered_lattice(21, n_states=300, avg_degree=8.0, vocab=64, width=8, span=4)
    theta = synth.label_scores(3, 64)
    masked = np.unique(l.label)[::7]
    masked = masked[(masked != BOS) & (masked != EOS)]
    for v in (-1e9, -3e38):
        th = theta.copy()
        th[masked] = v
        ref = theta.copy()
        ref[masked] = -np.inf
        r = ops.forward_backward(LatticeBatch.from_synth([l], device=dev), torch.from_numpy(th.astype(np.float32)))
        o, _ = oracle_fb(l, ref)
        assert np.isfinite(o["logZ"])
        assert abs(float(r.logz64[0]) - o["logZ"]) <= TOL
        assert np.max(np.abs(r.posterior.cpu().numpy() - o["posterior"])) <= 2e-6

@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_shapes_match_oracle(dev, seed):
    """Ragged batches of random shapes: chains, wide and narrow layers, high degrees (continuation
    pieces and wide groups), tiny lattices whose whole program is shorter than the rings."""
    rng = np.random.default_rng(100 + seed)
    V = int(rng.choice([96, 300]))
    lats = []
    for i in range(24):
        n = int(rng.choice([4, 5, 17, 60, 200, 700, 1500]))
        width = int(rng.choice([1, 2, 4, 16, 48] if V > 100 else [1, 2, 4, 16]))
        deg = float(rng.choice([1.5, 4.0, 10.0, 40.0] if V > 100 else [1.5, 4.0, 10.0]))
        lats.append(synth.layered_lattice(1000 * seed + i, n_states=n, avg_degree=min(deg, V - 4), vocab=V, width=width,
                                          span=int(rng.choice([1, 3, 8])), max_degree=min(V - 4, 64)))
    lats.append(synth._finish(2, V, [0], [EOS], [1]))
    theta = synth.label_scores(seed, V, mean=float(rng.choice([-2.0, 0.5])), std=1.0)
    for opts in (dict(), dict(group_mode=1), dict(group_mode=2, slots_per_lane=int(rng.choice([1, 2, 4])))):
        lat = LatticeBatch.from_synth(lats, device=dev, **opts)
        r = ops.forward_backward(lat, torch.from_numpy(theta))
        la, lb, post = r.logalpha.cpu().numpy(), r.logbeta.cpu().numpy(), r.posterior.cpu().numpy()
        for b, l in enumerate(lats):
            o, _ = oracle_fb(l, theta)
            r0, a0 = int(lat.row_off[b]), int(lat.arc_off[b])
            assert rec("random_shapes_logz", abs(float(r.logz64[b]) - o["logZ"])) <= TOL
            cmp_rows(la[r0:r0 + l.n_rows], o["logalpha"], tag="random_shapes_logalpha")
            cmp_rows(lb[r0:r0 + l.n_rows], o["logbeta"], tag="random_shapes_logbeta")
            # float32 rounding accumulates over the levels (up to 1500 here): 1e-5 instead of the
            # 2e-6 that holds at the BASELINE depth
            assert rec("random_shapes_post", np.max(np.abs(post[a0:a0 + l.n_arcs] - o["posterior"]))) <= 1e-5


def test_many_small_lattices_share_compute_units(dev):
    """More lattices than CUs: the 512- and 256-thread flavours (decoder loads for itself, two
    workgroups per CU) must give the same bits as the one-lattice-per-CU flavour."""
    V = 64
    base = [synth.layered_lattice(7000 + i, n_states=40 + 13 * (i % 9), avg_degree=5.0, vocab=V, width=1 + i % 7, span=3)
            for i in range(40)]
    theta = torch.from_numpy(synth.label_scores(9, V))
    ref = ops.forward_backward(LatticeBatch.from_synth(base, device=dev), theta)
    for reps in (8, 15):  # 320 and 600 lattices
        lat = LatticeBatch.from_synth(base * reps, device=dev)
        r = ops.forward_backward(lat, theta)
        rb = ops.backward(lat, theta)
        z = r.logz64.view(reps, len(base))
        assert torch.equal(z, ref.logz64.expand(reps, -1))
        assert torch.equal(rb.logz64, r.logz64)
        assert torch.equal(r.posterior.view(reps, -1), ref.posterior.expand(reps, -1))
        assert torch.equal(r.logbeta.view(reps, -1), ref.logbeta.expand(reps, -1))


def test_forward_backward_is_graph_capturable(dev):
    """The launch allocates nothing and keeps no state: after one warm-up call it can be captured
    in a HIP graph and replayed."""
    lats = _mixed_batch()
    lat = LatticeBatch.from_synth(lats, device=dev)
    theta = torch.from_numpy(synth.label_scores(7, 64)).to(dev)
    out = ops.forward_backward(lat, theta)           # warm-up: sets the LDS attribute, allocates outputs
    ref_z, ref_p = out.logz64.clone(), out.posterior.clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            out = ops.forward_backward(lat, theta, out=out)
    torch.cuda.current_stream().wait_stream(s)
    theta2 = theta.clone()
    for scale in (1.0, 0.5):
        theta.copy_(theta2 * scale)
        out.logz64.zero_(); out.posterior.zero_()
        g.replay()
        torch.cuda.synchronize()
        chk = ops.forward_backward(lat, theta)
        assert torch.equal(out.logz64, chk.logz64) and torch.equal(out.posterior, chk.posterior)
    theta.copy_(theta2)
    g.replay(); torch.cuda.synchronize()
    assert torch.equal(out.logz64, ref_z) and torch.equal(out.posterior, ref_p)

def test_repeated_launches_are_bit_identical(dev):
    """The loader / decoder / sweep waves synchronise through counters in LDS only: any ordering
    bug would show up as run-to-run differences.  Every flavour (one lattice per CU, two workgroups
    per CU with 512 and 256 threads, per-arc extras, wide groups) is launched many times."""
    theta = torch.from_numpy(synth.label_scores(1, 256)).to(dev)
    small = [synth.layered_lattice(9000 + i, n_states=30 + 17 * (i % 23), avg_degree=4.0 + (i % 5), vocab=256, width=1 + i % 9,
                                   span=1 + i % 4) for i in range(700)]
    cases = [(synth.bench_batch(32), dict()), (small[:300], dict()), (small, dict()), (small, dict(extras=True)),
             (synth.bench_batch(16), dict(group_mode=2))]
    for lats, kw in cases:
        extras = kw.pop("extras", False)
        lat = LatticeBatch.from_synth(lats, device=dev, **kw)
        asc = torch.randn(lat.total_arcs, device=dev) * 0.2 if extras else None
        ref = ops.forward_backward(lat, theta, arc_scores=asc)
        refb = ops.backward(lat, theta, arc_scores=asc)
        for _ in range(60):
            r = ops.forward_backward(lat, theta, arc_scores=asc)
            assert torch.equal(r.logz64, ref.logz64) and torch.equal(r.posterior, ref.posterior)
            assert torch.equal(r.logalpha, ref.logalpha) and torch.equal(r.logbeta, ref.logbeta)
            assert torch.equal(ops.backward(lat, theta, arc_scores=asc).logz64, refb.logz64)

def test_large_vocabulary_uses_32_bit_records(dev):
    """More than 2046 marks do not fit the compact tile's 11-bit label field: the packer must fall
    back to 32-bit records (format code 4) and the results must not change."""
    V = 3000
    lats = [synth.layered_lattice(50 + i, n_states=400 + 90 * i, avg_degree=9.0, vocab=V, width=12, span=5) for i in range(3)]
    theta = synth.label_scores(6, V)
    lat = LatticeBatch.from_synth(lats, device=dev)
    assert set((lat.meta_host[:, _lib.META_FWD_U] & 0xFF).tolist()) <= {1, 2, 4}
    r = ops.forward_backward(lat, torch.from_numpy(theta))
    v = ops.viterbi(lat, torch.from_numpy(theta), pad=PAD)
    for b, l in enumerate(lats):
        o, _ = oracle_fb(l, theta)
        a0 = int(lat.arc_off[b])
        assert rec("large_vocab_logz", abs(float(r.logz64[b]) - o["logZ"])) <= TOL
        assert np.max(np.abs(r.posterior.cpu().numpy()[a0:a0 + l.n_arcs] - o["posterior"])) <= 2e-6
        best, path, arcs = O.viterbi(l.n_rows, l.src, l.label, l.dst, theta[l.label], 4000)
        n = int(v.lengths[b])
        assert n == len(path) and np.float32(best) == v.best.cpu().numpy()[b]
        assert np.array_equal(v.paths.cpu().numpy()[b, :n], path)

def test_fused_log_z_total(dev):
    """nfst_forward_backward can add every lattice's log Z into one of three rotating slots
    (the loss of a training step without a reduction kernel) and clears the next slot."""
    lats = _mixed_batch()
    lat = LatticeBatch.from_synth(lats, device=dev)
    total = torch.zeros(3, dtype=torch.float64, device=dev)
    out = None
    for step in range(7):
        theta = torch.from_numpy(synth.label_scores(step, 64)).to(dev)
        out = ops.forward_backward(lat, theta, out=out, total=total, total_slot=step % 3)
        want = float(out.logz64.sum())
        got = total.cpu().numpy()
        assert abs(got[step % 3] - want) <= 1e-9 * max(1.0, abs(want))
        assert got[(step + 1) % 3] == 0.0
    with pytest.raises(ValueError):
        ops.forward_backward(lat, theta, total=torch.zeros(3, device=dev))

def test_prepared_launch_follows_in_place_updates_of_the_scores(dev):
    """ops.ForwardBackwardLaunch: the host work of a step done once.  Every launch equals the wrapper's result on the scores
    as they are at that moment (updated in place, as an optimiser does), fills the rotating loss slot, and a different
    output set or a wrong `total` is refused as by the wrapper."""
    lats = _mixed_batch()
    lat = LatticeBatch.from_synth(lats, device=dev)
    theta = torch.from_numpy(synth.label_scores(0, 64)).to(dev)
    arc = torch.zeros(lat.total_arcs, device=dev)
    total = torch.zeros(3, dtype=torch.float64, device=dev)
    launch = ops.ForwardBackwardLaunch(lat, theta, arc_scores=arc, total=total)
    for step in range(5):
        theta.copy_(torch.from_numpy(synth.label_scores(step, 64)))
        arc.copy_(0.1 * torch.randn(lat.total_arcs, generator=torch.Generator().manual_seed(step)))
        got = launch(step % 3)
        want = ops.forward_backward(lat, theta.clone(), arc_scores=arc.clone())
        assert got is launch.out
        for a, b in zip(got, want):
            assert (a is None) == (b is None)
            if a is not None:
                assert torch.equal(a, b)
        t = total.cpu().numpy()
        assert abs(t[step % 3] - float(want.logz64.sum())) <= 1e-9 * max(1.0, abs(float(want.logz64.sum())))
        assert t[(step + 1) % 3] == 0.0
    with pytest.raises(ValueError):
        ops.ForwardBackwardLaunch(lat, theta, total=torch.zeros(3, device=dev))
    with pytest.raises(ValueError):
        ops.ForwardBackwardLaunch(lat, theta, want_posterior=False, out=launch.out)


def test_snips_shaped_batch(dev):
    """BASELINE configs[2]: a batch of 64 tagging-shaped lattices (S ~ 400..1500, V ~ 250, long and
    narrow: a few tag states per token position, up to ~750 positions)."""
    V = 250
    lats = synth.snips_shaped_batch(64, vocab=V)
    theta = synth.label_scores(64, V, mean=-1.5, std=0.8)
    lat = LatticeBatch.from_synth(lats, device=dev)
    assert int(lat.depth.max()) >= 500
    r = ops.forward_backward(lat, torch.from_numpy(theta))
    v = ops.viterbi(lat, torch.from_numpy(theta), pad=PAD)
    for b in range(0, 64, 3):
        l = lats[b]
        o, _ = oracle_fb(l, theta)
        r0, a0 = int(lat.row_off[b]), int(lat.arc_off[b])
        assert rec("snips_shaped_logz", abs(float(r.logz64[b]) - o["logZ"])) <= TOL  # configs[2]: 1e-5 absolute
        cmp_rows(r.logbeta.cpu().numpy()[r0:r0 + l.n_rows], o["logbeta"], tag="snips_shaped_logbeta")
        assert rec("snips_shaped_post", np.max(np.abs(r.posterior.cpu().numpy()[a0:a0 + l.n_arcs] - o["posterior"]))) <= 1e-5
        best, path, arcs = O.viterbi(l.n_rows, l.src, l.label, l.dst, theta[l.label], 4000)
        n = int(v.lengths[b])
        assert n == len(path) and np.float32(best) == v.best.cpu().numpy()[b]
        assert np.array_equal(v.paths.cpu().numpy()[b, :n], path)
    # every lattice: posteriors of the arcs leaving the start state sum to one
    post = r.posterior.cpu().numpy()
    for b, l in enumerate(lats):
        a0 = int(lat.arc_off[b])
        assert abs(post[a0:a0 + l.n_arcs][l.src == 0].sum() - 1.0) <= 1e-5


def test_sampling_and_viterbi_with_bf16_scores(dev):
    """BASELINE configs[4]: posterior sampling + Viterbi on a transliteration-shaped lattice with
    the scores rounded to bfloat16 (the engine accumulates in float32 either way).  Against the
    float32-score oracle: |log Z| difference <= 5e-2; against the oracle run on the
    rounded scores: the usual 1e-5 and bit-exact indices."""
    V = 64
    x = list(range(10, 22)); y = list(range(30, 41))
    l = synth.edit_lattice(x, y, vocab=V, seed=5)
    theta = synth.label_scores(12, V, mean=-1.0, std=0.7)
    theta_bf = torch.from_numpy(theta).to(torch.bfloat16).to(torch.float32).numpy()
    lat = LatticeBatch.from_synth([l], device=dev)
    o32, _ = oracle_fb(l, theta)
    obf, scbf = oracle_fb(l, theta_bf)
    r = ops.forward_backward(lat, torch.from_numpy(theta_bf))
    assert abs(float(r.logz64[0]) - obf["logZ"]) <= TOL
    assert abs(float(r.logz64[0]) - o32["logZ"]) <= 5e-2
    v = ops.viterbi(lat, torch.from_numpy(theta_bf), pad=PAD)
    best, path, arcs = O.viterbi(l.n_rows, l.src, l.label, l.dst, theta_bf[l.label], 4000)
    n = int(v.lengths[0])
    assert n == len(path) and np.float32(best) == v.best.cpu().numpy()[0]
    assert np.array_equal(v.paths.cpu().numpy()[0, :n], path) and np.array_equal(v.arcs.cpu().numpy()[0, :n], arcs)
    K, T = 16, int(lat.depth.max()) + 1
    u = np.random.default_rng(3).random((1, K, T)).astype(np.float32)
    s = ops.sample_paths(lat, torch.from_numpy(theta_bf), K, max_len=T, uniforms=torch.from_numpy(u), pad=PAD)
    ref = O.sample_paths(l.n_rows, l.src, l.label, l.dst, scbf, obf["logbeta"], u[0].astype(np.float64), PAD)
    safe = ref["margin"] > 1e-5
    assert safe.sum() >= K - 2
    assert np.array_equal(s.paths.cpu().numpy()[0][safe], ref["paths"][safe])
    assert np.max(np.abs(s.logq.cpu().numpy()[0][safe] - ref["logq"][safe])) <= 2e-5

@pytest.mark.parametrize("weighted", [False, True])
def test_fused_proposal_step(dev, weighted):
    """nfst_proposal_step against the numpy restatement of one Sampler.stateful_sample step
    (samplers.py:243-297): symbols and next states bit-exact away from CDF boundaries, log q and
    logsumexp within 2e-5; forced evaluation; with and without the beta-value gather."""
    V, K = 96, 24
    lats = [synth.layered_lattice(400 + i, n_states=60 + 35 * i, avg_degree=6.0, vocab=V, width=5, span=3, weighted=weighted)
            for i in range(5)]
    lat = LatticeBatch.from_synth(lats, device=dev)
    rng = np.random.default_rng(17)
    N = len(lats) * K
    # walkers somewhere on their lattice: random reachable states, consistent previous symbols
    state = np.zeros(N, np.int64); inp = np.full(N, BOS, np.int64)
    for b, l in enumerate(lats):
        for k in range(K):
            a = int(rng.integers(0, l.n_arcs))
            if k % 6 == 0:      # just after the implicit bos (transition[0, bos] = state 1)
                state[b * K + k], inp[b * K + k] = int(l.dst[(l.src == 0) & (l.label == BOS)][0]), BOS
            elif k % 6 == 1:    # finished: in the sink after eos / pad
                state[b * K + k], inp[b * K + k] = l.n_rows - 1, (EOS if k % 12 == 1 else PAD)
            else:
                state[b * K + k], inp[b * K + k] = l.dst[a], l.label[a]
    scores = rng.normal(0.0, 1.5, size=(N, V)).astype(np.float32)
    u = rng.random(N).astype(np.float32)
    values = rng.normal(0.0, 0.5, size=lat.total_rows).astype(np.float32)
    for temperature, use_values, has_to_end in ((1.0, False, False), (0.7, True, False), (1.3, False, True)):
        r = ops.proposal_step(lat, torch.from_numpy(state), torch.from_numpy(scores), k=K, inp=torch.from_numpy(inp),
                              values=torch.from_numpy(values) if use_values else None, pad=PAD, bos=BOS, eos=EOS,
                              has_to_end=has_to_end, temperature=temperature, uniforms=torch.from_numpy(u))
        sym, logq, logz, nxt = (x.cpu().numpy() for x in r)
        for b, l in enumerate(lats):
            em, tr = l.dense(weighted=weighted)
            sl = slice(b * K, (b + 1) * K)
            em_k = np.broadcast_to(em[None], (K,) + em.shape); tr_k = np.broadcast_to(tr[None], (K,) + tr.shape)
            r0 = int(lat.row_off[b])
            beta = np.broadcast_to(values[r0:r0 + l.n_rows][None], (K, l.n_rows)) if use_values else None
            length, max_length = (5, 3) if has_to_end else (2, 300)
            o = O.proposal_step(em_k, tr_k, scores[sl], inp[sl], state[sl], length, max_length, PAD, BOS, EOS,
                                temperature=temperature, beta=beta, uniforms=u[sl].astype(np.float64))
            finite = np.isfinite(o["logz"])  # a forced end (has_to_end) leaves walkers without an eos arc no legal mark
            assert finite.all() or has_to_end
            assert np.all(np.isneginf(logz[sl][~finite])) and np.all(np.isneginf(logq[sl][~finite]))
            assert np.max(np.abs(logz[sl][finite] - o["logz"][finite])) <= 2e-5
            safe = (o["margin"] > 1e-5) & finite
            assert safe.sum() > 0.9 * finite.sum()
            assert np.array_equal(sym[sl][safe], o["symbol"][safe]) and np.array_equal(nxt[sl][safe], o["next_state"][safe])
            assert np.max(np.abs(logq[sl][safe] - o["logq"][safe])) <= 2e-5
            # forced evaluation of the oracle's symbols
            f = ops.proposal_step(lat, torch.from_numpy(state), torch.from_numpy(scores), k=K, inp=torch.from_numpy(inp),
                                  values=torch.from_numpy(values) if use_values else None, pad=PAD, bos=BOS, eos=EOS,
                                  has_to_end=has_to_end, temperature=temperature,
                                  forced=torch.from_numpy(np.concatenate([sym[:b * K], o["symbol"], sym[(b + 1) * K:]])))
            assert np.max(np.abs(f.logq.cpu().numpy()[sl][finite] - o["logq"][finite])) <= 2e-5
            assert np.array_equal(f.next_state.cpu().numpy()[sl][finite], o["next_state"][finite])


def test_slot_ordered_extras_workspace_sequences(dev):
    """Per-arc extras travel to the sweeps through a per-batch workspace in tile-slot order, refilled
    by every launch unless only the (static) table weights are in play.  Any order of backward /
    forward_backward launches, with and without caller scores, must give the oracle's numbers."""
    lats = [synth.layered_lattice(s, n_states=120 + 40 * s, avg_degree=6.0, vocab=48, width=6, span=3, weighted=True) for s in range(3)]
    lat = LatticeBatch.from_synth(lats, device=dev)
    theta = synth.label_scores(2, 48)
    rng = np.random.default_rng(5)
    asc1 = rng.normal(0.0, 0.3, size=lat.total_arcs).astype(np.float32)
    asc2 = rng.normal(0.0, 0.3, size=lat.total_arcs).astype(np.float32)
    def want(asc):
        out = []
        for b, l in enumerate(lats):
            a0 = int(lat.arc_off[b])
            out.append(oracle_fb(l, theta, None if asc is None else asc[a0:a0 + l.n_arcs])[0]["logZ"])
        return np.array(out)
    th = torch.from_numpy(theta)
    plan = [("bwd", None), ("fb", None), ("bwd", None), ("fb", asc1), ("bwd", None), ("bwd", asc2), ("fb", None), ("fb", asc2),
            ("fb", None), ("bwd", asc1), ("fb", None)]
    for op, asc in plan:
        t = None if asc is None else torch.from_numpy(asc)
        r = ops.forward_backward(lat, th, arc_scores=t) if op == "fb" else ops.backward(lat, th, arc_scores=t)
        assert np.max(np.abs(r.logz64.cpu().numpy() - want(asc))) <= TOL, (op, asc is None)


# ----------------------------------------------------------------------------- neuralised beta
NEURAL = ["neural_layered12_h8", "neural_layered40_h16", "neural_layered90_h64", "neural_edit_h8", "neural_parallel_arcs_h8"]


def _neural_params(seed, V, H, scale=1.0):
    g = np.random.default_rng(seed)
    lim = np.sqrt(6.0 / (2 * H))  # xavier_uniform like scorers.py:958-967
    return dict(emb=g.standard_normal((V, H)).astype(np.float32), Wx=g.uniform(-lim, lim, (H, H)).astype(np.float32),
                Wh=(scale * g.uniform(-lim, lim, (H, H))).astype(np.float32),
                W=g.uniform(-np.sqrt(6.0 / (1 + H)), np.sqrt(6.0 / (1 + H)), (1, H)).astype(np.float32),
                bias=(0.3 * g.standard_normal(H)).astype(np.float32))


def _run_neural(lat, p):
    return ops.backward_neural(lat, *(torch.from_numpy(p[k]) for k in ("emb", "Wx", "Wh", "W", "bias")))


@pytest.mark.parametrize("name", NEURAL)
def test_beta_neural_matches_reference_fixture(dev, golden_dir, name):
    """compute_beta_per_sample with Wh != 0 (scorers.py:692-751), values produced by the reference."""
    with np.load(os.path.join(golden_dir, "beta_neural.npz")) as g:
        c = {k[len(name) + 1:]: g[k] for k in g.files if k.startswith(name + "_")}
    n_rows = int(c["n_rows"])
    lat = LatticeBatch.from_arcs([n_rows], [0, c["src"].shape[0]], c["src"], c["label"], c["dst"], c["emb"].shape[0], device=dev)
    r = _run_neural(lat, c)
    got = r.log_beta.cpu().numpy().astype(np.float64)
    ref = c["beta_per_sample"].astype(np.float64)
    np.testing.assert_allclose(np.exp(got), ref, rtol=5e-5)  # the reference itself is float32
    logb, bhat = O.beta_neural(n_rows, c["src"], c["label"], c["dst"], c["emb"], c["Wx"], c["Wh"], c["W"], c["bias"])
    assert np.max(np.abs(got - logb)) <= 2e-5
    assert np.max(np.abs(r.beta_hat.cpu().numpy() - bhat)) <= 2e-5


@pytest.mark.parametrize("H", [5, 8, 24, 32, 64, 100, 128, 192, 256, 320, 512])
def test_beta_neural_against_oracle(dev, H):
    """Mixed batch (sizes, a 200-way fan-out and fan-in, weighted tables) under every packing the
    host can choose: carry pieces and partial groups merge (beta, beta_hat) by weight."""
    V = 256
    src = [0] + [1] * 200 + list(range(2, 202)) + [202]
    lab = [BOS] + list(range(3, 203)) + [5] * 200 + [EOS]
    dst = [1] + list(range(2, 202)) + [202] * 200 + [203]
    star = synth._finish(204, V, src, lab, dst)
    lats = [star, synth.layered_lattice(51, n_states=150, avg_degree=6.0, vocab=V, width=4, span=3),
            synth.layered_lattice(52, n_states=60, avg_degree=3.0, vocab=V, width=1, span=2),
            synth.layered_lattice(53, n_states=300, avg_degree=12.0, vocab=V, width=24, span=2, max_degree=40)]
    p = _neural_params(H, V, H, scale=2.0)
    refs = [O.beta_neural(l.n_rows, l.src, l.label, l.dst, p["emb"], p["Wx"], p["Wh"], p["W"], p["bias"]) for l in lats]
    all_opts = (dict(), dict(slots_per_lane=1), dict(slots_per_lane=2, group_mode=1), dict(slots_per_lane=4, no_compact=True),
                dict(group_mode=2)) if H in (8, 256) else (dict(), dict(group_mode=1, slots_per_lane=1))
    for opts in all_opts:
        lat = LatticeBatch.from_synth(lats, device=dev, **opts)
        r = _run_neural(lat, p)
        again = _run_neural(lat, p)  # the phases hand rows over through LDS and L2: no race, no stale line
        assert torch.equal(r.log_beta, again.log_beta) and torch.equal(r.beta_hat, again.beta_hat)
        lb, bh = r.log_beta.cpu().numpy().astype(np.float64), r.beta_hat.cpu().numpy()
        for b, (l, (logb, bhat)) in enumerate(zip(lats, refs)):
            r0 = int(lat.row_off[b])
            cmp_rows(lb[r0:r0 + l.n_rows], logb, tol=3e-5)
            assert np.max(np.abs(bh[r0:r0 + l.n_rows] - bhat)) <= 3e-5, (H, opts, b)


def test_beta_neural_weighted_tables_and_zero_wh(dev):
    """Weighted tables add their arc weight to the compatibility score; with Wh = 0 the sweep is
    the plain beta sweep with theta[l] = W . tanh(Wx e(l) + bias)."""
    V, H = 64, 16
    lats = [synth.layered_lattice(61 + i, n_states=80 + 40 * i, avg_degree=5.0, vocab=V, width=3, span=3, weighted=True) for i in range(3)]
    p = _neural_params(5, V, H)
    p["Wh"][:] = 0
    lat = LatticeBatch.from_synth(lats, device=dev)
    r = _run_neural(lat, p)
    theta = (np.tanh(p["emb"].astype(np.float64) @ p["Wx"].T.astype(np.float64) + p["bias"]) @ p["W"].reshape(-1).astype(np.float64))
    plain = ops.backward(lat, torch.from_numpy(theta.astype(np.float32)))
    assert torch.max(torch.abs(r.log_beta - plain.logbeta)[torch.isfinite(plain.logbeta)]) <= 2e-5
    for b, l in enumerate(lats):
        o, _ = oracle_fb(l, theta)
        r0 = int(lat.row_off[b])
        cmp_rows(r.log_beta.cpu().numpy()[r0:r0 + l.n_rows], o["logbeta"], tol=2e-5)
    with pytest.raises(ValueError):
        ops.backward_neural(lat, torch.zeros(V, H), torch.zeros(H, H), torch.zeros(H + 1, H), torch.zeros(H), torch.zeros(H))
    # table weights together with the state term
    p = _neural_params(6, V, H, scale=2.0)
    r = _run_neural(lat, p)
    for b, l in enumerate(lats):
        logb, bhat = O.beta_neural(l.n_rows, l.src, l.label, l.dst, p["emb"], p["Wx"], p["Wh"], p["W"], p["bias"], arc_w=l.weight)
        r0 = int(lat.row_off[b])
        cmp_rows(r.log_beta.cpu().numpy()[r0:r0 + l.n_rows], logb, tol=3e-5)
        assert np.max(np.abs(r.beta_hat.cpu().numpy()[r0:r0 + l.n_rows] - bhat)) <= 3e-5


# ----------------------------------------------------------------------------- gradient of the neuralised beta
NEURAL_GRAD = ["grad_layered12_h8", "grad_layered40_h16", "grad_layered60_h64", "grad_edit_h8"]
PARAMS = ("emb", "Wx", "Wh", "W", "bias")


def _neural_grads(lat, p, coef_rows, coef_hat=None):
    """loss = sum coef . log beta (+ sum coef_hat . beta_hat) through ops.backward_neural; returns
    (loss, NeuralBeta, {name: grad})."""
    t = {k: torch.from_numpy(np.asarray(p[k], np.float32)).to(lat.device).requires_grad_(True) for k in PARAMS}
    r = ops.backward_neural(lat, *(t[k] for k in PARAMS))
    c = torch.from_numpy(np.asarray(coef_rows, np.float32)).to(lat.device)
    loss = (c[c != 0] * r.log_beta[c != 0]).sum()
    if coef_hat is not None:
        loss = loss + (torch.from_numpy(np.asarray(coef_hat, np.float32)).to(lat.device) * r.beta_hat).sum()
    loss.backward()
    return float(loss.detach()), r, {k: t[k].grad.detach().cpu().numpy().astype(np.float64) for k in PARAMS}


def _rel(got, ref):
    return float(np.max(np.abs(got - ref)) / max(np.max(np.abs(ref)), 1e-30))


@pytest.mark.parametrize("name", NEURAL_GRAD)
def test_beta_neural_grad_matches_reference_differences(dev, golden_dir, name):
    """d/d(Wh, Wx, W, beta_bias, embeddings) of sum coef . log beta (tune_proposal trains these through
    compute_beta, lightning.py:339-406) against central differences of the REFERENCE's own forward pass
    in float64 (torch.autograd refuses the reference's in-place updates) and against float64 autograd
    over the restatement."""
    with np.load(os.path.join(golden_dir, "beta_neural_grad.npz")) as g:
        c = {k[len(name) + 1:]: g[k] for k in g.files if k.startswith(name + "_")}
    n_rows = int(c["n_rows"])
    lat = LatticeBatch.from_arcs([n_rows], [0, c["src"].shape[0]], c["src"], c["label"], c["dst"], c["emb"].shape[0], device=dev)
    loss, _, got = _neural_grads(lat, c, c["coef"])
    assert abs(loss - float(c["loss"])) <= 2e-5 * max(1.0, abs(float(c["loss"])))
    for k in PARAMS:
        scale = max(1.0, float(np.max(np.abs(c["dd_" + k]))))
        for d, dd in zip(c["dir_" + k], c["dd_" + k]):
            val = float((got[k].reshape(-1) * d.astype(np.float64).reshape(-1)).sum())
            assert rec("neural_grad_vs_reference_fd", abs(val - dd) / scale) <= 1e-4, (k, val, dd)
    _, _, _, ref = O.beta_neural_grad(n_rows, c["src"], c["label"], c["dst"], c["emb"], c["Wx"], c["Wh"], c["W"], c["bias"], c["coef"])
    for k in PARAMS:
        assert rec("neural_grad_rel", _rel(got[k].reshape(-1), ref[k].reshape(-1))) <= 1e-4, k


@pytest.mark.parametrize("H", [5, 8, 16, 24, 32, 64, 100, 128, 192, 256, 384, 512])
def test_beta_neural_grad_against_autograd(dev, H):
    """Mixed batch (weighted tables; a near-sequential, a layered and a wide lattice) under several
    packings, with gradients entering through log beta AND beta_hat: float64 autograd over the
    restatement, <= 1e-4 relative."""
    V = 64
    lats = [synth.layered_lattice(71, n_states=120, avg_degree=5.0, vocab=V, width=4, span=3, weighted=True),
            synth.layered_lattice(72, n_states=40, avg_degree=3.0, vocab=V, width=1, span=2, weighted=True),
            synth.layered_lattice(73, n_states=200, avg_degree=10.0, vocab=V, width=20, span=2, max_degree=30, weighted=True)]
    p = _neural_params(H + 1, V, H, scale=2.0)
    rng = np.random.default_rng(H)
    coefs = [rng.normal(size=l.n_rows) for l in lats]
    chats = [0.3 * rng.normal(size=(l.n_rows, H)) for l in lats]
    ref = {k: 0.0 for k in PARAMS}
    for l, c, ch in zip(lats, coefs, chats):
        _, logb, _, g = O.beta_neural_grad(l.n_rows, l.src, l.label, l.dst, p["emb"], p["Wx"], p["Wh"], p["W"], p["bias"], c,
                                           arc_w=l.weight, coef_hat=ch)
        assert np.all(np.isfinite(logb[np.asarray(c) != 0]))
        for k in PARAMS:
            ref[k] = ref[k] + g[k].reshape(np.shape(p[k]))
    all_opts = (dict(), dict(slots_per_lane=1), dict(slots_per_lane=2, group_mode=1), dict(slots_per_lane=4, no_compact=True),
                dict(group_mode=2)) if H in (8, 256) else (dict(), dict(group_mode=1, slots_per_lane=1))
    for opts in all_opts:
        lat = LatticeBatch.from_synth(lats, device=dev, **opts)
        rows = np.zeros(lat.total_rows)
        hat = np.zeros((lat.total_rows, H))
        for b, (l, c, ch) in enumerate(zip(lats, coefs, chats)):
            r0 = int(lat.row_off[b])
            rows[r0:r0 + l.n_rows] = c
            hat[r0:r0 + l.n_rows] = ch
        _, _, got = _neural_grads(lat, p, rows, hat)
        for k in PARAMS:
            assert rec(f"neural_grad_rel_H{H}", _rel(got[k], ref[k])) <= 1e-4, (H, opts, k)


def test_beta_neural_grad_funnels(dev):
    """Fan-in / fan-out of 200 arcs twice in a row (carry pieces, partial groups whose scratch rows are
    rewritten at the next level), forward values and gradients."""
    V, H = 256, 16
    src = [0] + [1] * 200 + list(range(2, 202)) + [202] * 200 + list(range(203, 403)) + [403]
    lab = [BOS] + list(range(3, 203)) + [5] * 200 + list(range(3, 203)) + [6] * 200 + [EOS]
    dst = [1] + list(range(2, 202)) + [202] * 200 + list(range(203, 403)) + [403] * 200 + [404]
    l = synth._finish(405, V, src, lab, dst)
    p = _neural_params(9, V, H, scale=2.0)
    coef = np.random.default_rng(3).normal(size=l.n_rows)
    _, logb, bhat, ref = O.beta_neural_grad(l.n_rows, l.src, l.label, l.dst, p["emb"], p["Wx"], p["Wh"], p["W"], p["bias"], coef)
    for opts in (dict(), dict(slots_per_lane=1), dict(group_mode=2), dict(slots_per_lane=2, group_mode=1)):
        lat = LatticeBatch.from_synth([l], device=dev, **opts)
        _, r, got = _neural_grads(lat, p, coef)
        cmp_rows(r.log_beta.detach().cpu().numpy()[: l.n_rows], logb, tol=3e-5)
        assert np.max(np.abs(r.beta_hat.detach().cpu().numpy()[: l.n_rows] - bhat)) <= 3e-5
        for k in PARAMS:
            assert rec("neural_grad_rel_funnels", _rel(got[k], ref[k].reshape(np.shape(p[k])))) <= 1e-4, (opts, k)


# ----------------------------------------------------------------------------- tile waves (one lattice per CU)
def test_tile_waves_short_programs_and_flavour_bits(dev):
    """The tile-wave kernels (all-compact batches, at most one lattice per CU) on programs shorter than the
    four tile waves of a sweep, programs of 1 .. 9 tiles, wide groups, table weights + caller scores, and
    beside them the loader / decoder / sweep pipeline (switch tw = 0): same bits for log Z, the row outputs and
    the posteriors, oracle parity for both, beta-only sweep included."""
    V = 40
    lats = []
    for n in (1, 2, 3, 4, 5, 7, 9, 13):  # chains of n levels: n tiles per direction
        lats.append(synth._finish(n + 1, V, list(range(n)), [3 + (i % 30) for i in range(n)], list(range(1, n + 1))))
    lats += [synth.layered_lattice(90 + i, n_states=30 + 25 * i, avg_degree=3.0 + i, vocab=V, width=1 + i, span=1 + i % 3, weighted=True)
             for i in range(6)]
    lats.append(synth.layered_lattice(97, n_states=300, avg_degree=12.0, vocab=V, width=24, span=2, max_degree=36, weighted=True))
    theta = synth.label_scores(7, V)
    th = torch.from_numpy(theta)
    for weighted_ok, opts in ((False, dict()), (True, dict()), (True, dict(group_mode=2))):
        import dataclasses
        use = [l for l in lats if l.weight is None] if not weighted_ok else \
            [l if l.weight is not None else dataclasses.replace(l, weight=np.zeros(l.n_arcs, np.float32)) for l in lats]
        lat = LatticeBatch.from_synth(use, device=dev, **opts)
        assert lat._h["reserved0"] & 1, "every program of this batch should be compact"
        asc = np.random.default_rng(len(use)).normal(0.0, 0.4, size=lat.total_arcs).astype(np.float32)
        for scores in (None, asc):
            t = None if scores is None else torch.from_numpy(scores)
            res = {}
            for tw in (1, 0):
                with _lib.tuning(tw=tw):
                    res[tw] = (ops.forward_backward(lat, th, arc_scores=t), ops.backward(lat, th, arc_scores=t))
            (fa, ba), (fb, bb) = res[1], res[0]
            assert torch.equal(fa.logz64, fb.logz64) and torch.equal(fa.logalpha, fb.logalpha) and torch.equal(fa.logbeta, fb.logbeta)
            assert torch.equal(ba.logz64, bb.logz64) and torch.equal(ba.logbeta, bb.logbeta)
            assert float(torch.max(torch.abs(fa.posterior - fb.posterior))) <= 2e-7  # (the posterior pass is the same code)
            for b, l in enumerate(use):
                a0 = int(lat.arc_off[b])
                o, _ = oracle_fb(l, theta, None if scores is None else scores[a0:a0 + l.n_arcs])
                assert rec("tile_waves_logz", abs(float(fa.logz64[b]) - o["logZ"])) <= TOL, (weighted_ok, opts, scores is None, b, l.n_rows)
                assert abs(float(ba.logz64[b]) - o["logZ"]) <= TOL
                assert np.max(np.abs(fa.posterior.cpu().numpy()[a0:a0 + l.n_arcs] - o["posterior"])) <= 2e-6
