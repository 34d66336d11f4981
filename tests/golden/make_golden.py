#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REAL reference.

Runs only in the development container (the reference lives at /root/reference
and never travels to the GPU box).  It imports the reference's own hot-path
modules on CPU -- with inert placeholder modules for the three third-party
packages that are not installed (bidict, mfst, pynini) --
feeds them seeded synthetic lattices from ``nfst_amd.synth`` and stores inputs
and the reference's outputs as small ``.npz`` files (plain arrays, no pickles).

    python tests/golden/make_golden.py

What is pinned (reference file:line):
  beta_*.npz     FSAGRUScorer.compute_beta_per_sample / compute_beta_parallel
                 (scorers.py:692-751, 753-856) with Wh = 0, i.e. arc weight
                 exp(theta[label]);  includes the parallel-arc quirk fixture.
  beta_neural.npz  the same two functions with Wh != 0 (Tree-LSTM-style messages).
  beta_neural_grad.npz  directional derivatives of compute_beta_per_sample w.r.t. Wh, Wx, W,
                 beta_bias and the embeddings (central differences of the reference's forward pass in
                 float64 -- torch.autograd refuses the reference's in-place updates; tune_proposal,
                 lightning.py:339-406).
  gather.npz     set_masks/set_k (877-918), update_fsa_state (683-690),
                 mask_out_invalid (1037-1054 on top of 314-338) on a collated,
                 pad-padded batch (dataset_reader.py:175-186).
  sampler.npz    Sampler.sample / stateful_sample / stripping_pad
                 (samplers.py:137-335) with the FSAMaskScorer proposal.  (The
                 reference's forced-scoring mode, to_evaluate != None, raises at
                 samplers.py:318 -- torch.stack of an empty list -- so it cannot
                 be pinned.)
  iwae.npz       Estimators.iwae (estimatros.py:11-44) + WFSTScorer.wfst_score
                 (scorers.py:1671-1687).
  evalseq.npz    StaticRNNScorer.evaluate_seq_with_temp arithmetic
                 (scorers.py:1530-1614, masks 89-134, smoothing 1502-1528) on
                 supplied score tensors (the RNN itself is out of scope).
  evalseq_grad.npz  the same function's gradient with respect to the supplied scores
                 (torch.autograd through the reference's code; lightning.py:511-516
                 trains p~ through it).
  gpt2.npz       GPT2Wrapper.forward (transformer.py:34-52) on supplied logits: value
                 and gradient.
  strip.npz      Sampler.stripping_pad (samplers.py:162-180) with pad = 0 and pad = 7.
  sampler_beta.npz  Sampler.stateful_sample with FSAGRUScorer(use_beta=True): recorded prefix
                 scores, beta, insertion / length penalties (scorers.py:577-601, 630-681).
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)


def _install_placeholders():
    class _bidict(dict):
        @property
        def inverse(self):
            return {v: k for k, v in self.items()}

    m = types.ModuleType("bidict")
    m.bidict = _bidict
    sys.modules["bidict"] = m

    class _Inert:
        def __init__(self, *a, **k):
            pass

        def create_from_string(self, *a, **k):
            return self

    m = types.ModuleType("mfst")
    m.FST = _Inert
    m.AbstractSemiringWeight = type("AbstractSemiringWeight", (), {})
    m.BooleanSemiringWeight = type("BooleanSemiringWeight", (), {})
    sys.modules["mfst"] = m
    m = types.ModuleType("pynini")
    m.Fst = object
    m.Weight = object
    m.Arc = object
    sys.modules["pynini"] = m


_install_placeholders()
sys.path.insert(0, "/root/reference")

import torch  # noqa: E402

from src.util.preprocess_util import Vocab  # noqa: E402
from src.modules.scorers import (  # noqa: E402
    FSAGRUScorer,
    FSAMaskScorer,
    WFSTScorer,
    StaticRNNScorer,
    CompositeScorer,
)
from src.modules.samplers import Sampler  # noqa: E402
from src.modules.estimatros import Estimators  # noqa: E402

from nfst_amd import synth  # noqa: E402

torch.set_num_threads(4)
PAD, BOS, EOS = synth.PAD, synth.BOS, synth.EOS

Vocab.set_non_reserved_offset(3)
for w in ("input-mark", "output-mark", "insertion-mark"):
    Vocab.add_word(w)  # -> ids 3, 4, 5
assert Vocab.lookup("insertion-mark") == 5


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: " + ", ".join(f"{k}{list(np.shape(v))}" for k, v in arrays.items()))


def beta_scorer(V, H=8, seed=0):
    torch.manual_seed(seed)
    sc = FSAGRUScorer(hid_dim=H, vocab_size=V, pad=PAD, bos=BOS, eos=EOS, use_beta=True, max_length=64)
    sc.eval()
    with torch.no_grad():
        sc.Wh.zero_()  # arc weight depends on the label only
        sc.beta_bias.copy_(0.3 * torch.randn(H))
        sc.W.copy_(torch.randn(1, H))
        emb = sc.embeddings.weight  # [V, H]
        theta = (sc.W @ torch.tanh(sc.Wx @ emb.t() + sc.beta_bias[:, None])).reshape(-1)
    sc._probe_weights = {
        "emb": sc.embeddings.weight.detach().numpy().astype(np.float32),
        "Wx": sc.Wx.detach().numpy().astype(np.float32),
        "Wh": sc.Wh.detach().numpy().astype(np.float32),
        "W": sc.W.detach().numpy().astype(np.float32),
        "bias": sc.beta_bias.detach().numpy().astype(np.float32),
    }
    return sc, theta.detach().numpy().astype(np.float32)


def lattice_arrays(lat, prefix=""):
    em, tr = lat.dense()
    return {
        prefix + "emission": em,
        prefix + "transition": tr,
        prefix + "src": lat.src,
        prefix + "label": lat.label,
        prefix + "dst": lat.dst,
    }


def make_beta():
    cases = {
        "beta_layered12": synth.layered_lattice(11, n_states=12, avg_degree=3.0, vocab=47, width=3, span=2),
        "beta_layered40": synth.layered_lattice(12, n_states=40, avg_degree=4.0, vocab=47, width=4, span=3),
        "beta_layered120": synth.layered_lattice(13, n_states=120, avg_degree=6.0, vocab=47, width=6, span=4),
        "beta_edit": synth.edit_lattice([10, 11, 12], [20, 21, 22, 23], vocab=47, seed=5),
    }
    for name, lat in cases.items():
        sc, theta = beta_scorer(lat.vocab, seed=len(name))
        em, tr = lat.dense()
        tr_t = torch.from_numpy(tr)
        with torch.no_grad():
            b_serial = sc.compute_beta_per_sample(tr_t).numpy().astype(np.float32)
            sc.set_masks(emission=torch.from_numpy(em)[None], transition=tr_t[None])
            sc.set_k(2)
            b_par = sc.compute_beta().numpy().astype(np.float32)
        assert b_par.shape[0] == 2 and np.array_equal(b_par[0], b_par[1])
        save(
            name + ".npz",
            theta=theta,
            beta_per_sample=b_serial,
            beta_parallel=b_par[0],
            **sc._probe_weights,
            **lattice_arrays(lat),
        )

    # The parallel-arc quirk: two labels between one state pair.
    # states: 0 -bos-> 1 ; 1 -{6,7}-> 2 (parallel) ; 1 -8-> 3 ; 2 -9-> 4 ; 3 -10-> 4 ; 4 -eos-> 5(sink)
    V = 16
    src = [0, 1, 1, 1, 2, 3, 4]
    lab = [BOS, 6, 7, 8, 9, 10, EOS]
    dst = [1, 2, 2, 3, 4, 4, 5]
    lat = synth._finish(6, V, src, lab, dst)
    sc, theta = beta_scorer(V, seed=3)
    em, tr = lat.dense()
    with torch.no_grad():
        b_serial = sc.compute_beta_per_sample(torch.from_numpy(tr)).numpy().astype(np.float32)
        sc.set_masks(emission=torch.from_numpy(em)[None], transition=torch.from_numpy(tr)[None])
        sc.set_k(1)
        b_par = sc.compute_beta().numpy().astype(np.float32)[0]
    save("beta_parallel_arc_quirk.npz", theta=theta, beta_per_sample=b_serial, beta_parallel=b_par, **sc._probe_weights, **lattice_arrays(lat))


def make_beta_neural():
    """Wh != 0: the Tree-LSTM-style messages of compute_beta_per_sample (scorers.py:692-751), and
    compute_beta_parallel (753-856) on the same lattices (none has parallel arcs between a state pair
    except where noted, so both agree)."""
    cases = {
        "neural_layered12_h8": (synth.layered_lattice(31, n_states=12, avg_degree=3.0, vocab=29, width=3, span=2), 8),
        "neural_layered40_h16": (synth.layered_lattice(32, n_states=40, avg_degree=4.0, vocab=29, width=4, span=3), 16),
        "neural_layered90_h64": (synth.layered_lattice(33, n_states=90, avg_degree=5.0, vocab=29, width=5, span=3), 64),
        "neural_edit_h8": (synth.edit_lattice([10, 11, 12], [20, 21, 22, 23], vocab=29, seed=6), 8),
        "neural_parallel_arcs_h8": (synth.layered_lattice(34, n_states=20, avg_degree=4.0, vocab=29, width=2, span=2), 8),
    }
    out = {}
    for name, (lat, H) in cases.items():
        torch.manual_seed(len(name) + H)
        sc = FSAGRUScorer(hid_dim=H, vocab_size=lat.vocab, pad=PAD, bos=BOS, eos=EOS, use_beta=True, max_length=64)
        sc.eval()
        with torch.no_grad():
            sc.beta_bias.copy_(0.3 * torch.randn(H))
            sc.Wh.mul_(2.0)  # make the state term matter
        em, tr = lat.dense()
        tr_t = torch.from_numpy(tr)
        with torch.no_grad():
            b_serial = sc.compute_beta_per_sample(tr_t).numpy().astype(np.float32)
            sc.set_masks(emission=torch.from_numpy(em)[None], transition=tr_t[None])
            sc.set_k(1)
            b_par = sc.compute_beta().numpy().astype(np.float32)[0]
        out.update({f"{name}_{k}": v for k, v in dict(
            beta_per_sample=b_serial, beta_parallel=b_par,
            emb=sc.embeddings.weight.detach().numpy().astype(np.float32), Wx=sc.Wx.detach().numpy().astype(np.float32),
            Wh=sc.Wh.detach().numpy().astype(np.float32), W=sc.W.detach().numpy().astype(np.float32),
            bias=sc.beta_bias.detach().numpy().astype(np.float32), n_rows=np.int64(lat.n_rows),
            src=lat.src, label=lat.label, dst=lat.dst, transition=tr).items()})
    save("beta_neural.npz", **out)


def make_beta_neural_grad():
    """Derivatives of compute_beta_per_sample (scorers.py:692-751) with respect to the parameters
    tune_proposal trains through compute_beta (lightning.py:339-406; scorers.py:954-970).  The
    reference's two beta functions update their tensors in place (scorers.py:744-747, 806-817) and
    torch.autograd refuses to differentiate either ("modified by an inplace operation"), so the
    fixture holds DIRECTIONAL derivatives of the reference's own forward pass instead: central
    differences in float64 of L = sum_s coef[s] log beta(s) (states reachable from state 0) along
    random directions in every parameter."""
    cases = {
        "grad_layered12_h8": (synth.layered_lattice(41, n_states=12, avg_degree=3.0, vocab=29, width=3, span=2), 8),
        "grad_layered40_h16": (synth.layered_lattice(42, n_states=40, avg_degree=4.0, vocab=29, width=4, span=3), 16),
        "grad_layered60_h64": (synth.layered_lattice(43, n_states=60, avg_degree=5.0, vocab=29, width=5, span=3), 64),
        "grad_edit_h8": (synth.edit_lattice([10, 11, 12], [20, 21, 22, 23], vocab=29, seed=7), 8),
    }
    out = {}
    torch.set_default_dtype(torch.float64)
    try:
        for name, (lat, H) in cases.items():
            torch.manual_seed(len(name) + H)
            sc = FSAGRUScorer(hid_dim=H, vocab_size=lat.vocab, pad=PAD, bos=BOS, eos=EOS, use_beta=True, max_length=64)
            sc.eval()
            with torch.no_grad():
                sc.beta_bias.copy_(0.3 * torch.randn(H))
                sc.Wh.mul_(2.0)
                for p in sc.parameters():  # parameters that are exactly representable in float32
                    p.copy_(p.float().double())
            _, tr = lat.dense()
            tr_t = torch.from_numpy(tr)
            reach = np.zeros(lat.n_rows, bool)
            reach[0] = True
            for _ in range(lat.n_rows):
                reach[lat.dst[reach[lat.src]]] = True
            rng = np.random.default_rng(len(name))
            coef = np.where(reach, rng.normal(size=lat.n_rows), 0.0).astype(np.float32)
            coef_t, reach_t = torch.from_numpy(coef.astype(np.float64)), torch.from_numpy(reach)

            def loss():
                with torch.no_grad():
                    beta = sc.compute_beta_per_sample(tr_t)[: lat.n_rows]
                return float((coef_t[reach_t] * torch.log(beta[reach_t])).sum())

            params = dict(emb=sc.embeddings.weight, Wx=sc.Wx, Wh=sc.Wh, W=sc.W, bias=sc.beta_bias)
            rec = dict(coef=coef, loss=np.float64(loss()), n_rows=np.int64(lat.n_rows), src=lat.src, label=lat.label, dst=lat.dst)
            for k, p in params.items():
                rec[k] = p.detach().numpy().astype(np.float32)
            eps = 1e-5
            for k, p in params.items():
                dirs, dds = [], []
                for _ in range(3):
                    d = torch.from_numpy(rng.normal(size=tuple(p.shape)))
                    with torch.no_grad():
                        p.add_(eps * d); up = loss()
                        p.sub_(2 * eps * d); dn = loss()
                        p.add_(eps * d)
                    dirs.append(d.numpy().astype(np.float32))
                    dds.append((up - dn) / (2 * eps))
                # (the stored direction is the float32 rounding of the one used: 6e-8 relative)
                rec[f"dir_{k}"] = np.stack(dirs)
                rec[f"dd_{k}"] = np.asarray(dds, np.float64)
            out.update({f"{name}_{k}": v for k, v in rec.items()})
    finally:
        torch.set_default_dtype(torch.float32)
    save("beta_neural_grad.npz", **out)


def make_gather():
    V = 24
    lats = [
        synth.layered_lattice(21, n_states=18, avg_degree=3.0, vocab=V, width=3, span=2),
        synth.layered_lattice(22, n_states=30, avg_degree=4.0, vocab=V, width=4, span=3),
        synth.edit_lattice([10, 11], [12, 13, 14], vocab=V, seed=9),
    ]
    rng = np.random.default_rng(77)
    out = {}
    for tag, pad_id in (("pad0", 0), ("pad7", 7)):
        em, tr = synth.collate_dense([l.dense() for l in lats], pad=pad_id)
        K = 3
        sc = FSAMaskScorer(hid_dim=4, vocab_size=V, pad=PAD, bos=BOS, eos=EOS, max_length=20)
        sc.set_masks(emission=torch.from_numpy(em), transition=torch.from_numpy(tr))
        sc.set_k(K)
        assert sc.transition_k.shape == (len(lats) * K, em.shape[1], V)
        N = len(lats) * K
        # random (state, label) pairs on real rows of each lattice (row < n_rows)
        reps = 40
        states = np.zeros((reps, N), dtype=np.int64)
        labels = rng.integers(0, V, size=(reps, N)).astype(np.int64)
        for n in range(N):
            states[:, n] = rng.integers(0, lats[n // K].n_rows, size=reps)
        nxt = np.zeros_like(states)
        masks_short = np.zeros((reps, N, V), dtype=np.float32)
        masks_long = np.zeros((reps, N, V), dtype=np.float32)
        for r in range(reps):
            st = torch.from_numpy(states[r])
            lb = torch.from_numpy(labels[r])
            nxt[r] = sc.update_fsa_state(lb, st).numpy()
            masks_short[r] = sc.mask_out_invalid(lb, {"state": st, "length": 5}).numpy()
            masks_long[r] = sc.mask_out_invalid(lb, {"state": st, "length": 21}).numpy()
        out.update(
            {
                f"{tag}_emission": em,
                f"{tag}_transition": tr,
                f"{tag}_states": states,
                f"{tag}_labels": labels,
                f"{tag}_next": nxt,
                f"{tag}_mask_len5": masks_short,
                f"{tag}_mask_len21": masks_long,
            }
        )
    # float emission tables (weighted machines, scorers.py:1011-1027): mask_out_invalid adds the state's row of log
    # weights (1049-1053); and the beta-logit gather of the use_beta proposal (scorers.py:581-593) on the same tables
    wl = [synth.layered_lattice(23 + i, n_states=16 + 9 * i, avg_degree=3.0, vocab=V, width=3, span=2, weighted=True) for i in range(3)]
    em, tr = synth.collate_dense([l.dense(weighted=True) for l in wl], pad=0)
    K = 3
    sc = FSAMaskScorer(hid_dim=4, vocab_size=V, pad=PAD, bos=BOS, eos=EOS, max_length=20)
    sc.set_masks(emission=torch.from_numpy(em), transition=torch.from_numpy(tr))
    sc.set_k(K)
    N, reps = len(wl) * K, 30
    states = np.zeros((reps, N), dtype=np.int64)
    labels = rng.integers(0, V, size=(reps, N)).astype(np.int64)
    for n in range(N):
        states[:, n] = rng.integers(0, wl[n // K].n_rows, size=reps)
    beta = rng.normal(size=(N, em.shape[1])).astype(np.float32)
    masks = np.zeros((reps, N, V), dtype=np.float32)
    blog = np.zeros((reps, N, V), dtype=np.float32)
    for r in range(reps):
        st = torch.from_numpy(states[r])
        masks[r] = sc.mask_out_invalid(torch.from_numpy(labels[r]), {"state": st, "length": 5}).numpy()
        transition = sc.transition_k[torch.arange(st.shape[0]), st]          # scorers.py:584-588
        blog[r] = torch.gather(torch.from_numpy(beta), 1, transition).numpy()  # :589
    out.update(w_emission=em, w_transition=tr, w_states=states, w_labels=labels, w_mask_len5=masks, w_beta=beta, w_beta_logits=blog,
               w_n_rows=np.array([l.n_rows for l in wl], dtype=np.int64))
    out["K"] = np.int64(3)
    out["max_length"] = np.int64(20)
    out["n_rows"] = np.array([l.n_rows for l in lats], dtype=np.int64)
    save("gather.npz", **out)


def make_sampler_and_iwae():
    V = 24
    lats = [
        synth.layered_lattice(31, n_states=20, avg_degree=3.0, vocab=V, width=3, span=2),
        synth.edit_lattice([10, 11, 12], [13, 14], vocab=V, seed=4),
    ]
    em, tr = synth.collate_dense([l.dense() for l in lats], pad=PAD)
    B, K = len(lats), 8
    sc = FSAMaskScorer(hid_dim=4, vocab_size=V, pad=PAD, bos=BOS, eos=EOS, max_length=48)
    sampler = Sampler(sc)
    sampler.set_masks(transition=torch.from_numpy(tr), emission=torch.from_numpy(em))
    sampler.set_k(K)
    torch.manual_seed(1234)
    with torch.no_grad():
        log_q, samples = sampler.sample(B * K)
        stripped = sampler.stripping_pad(samples)
    save(
        "sampler.npz",
        emission=em,
        transition=tr,
        K=np.int64(K),
        max_length=np.int64(48),
        log_q=log_q.numpy().astype(np.float32),
        samples=samples.numpy(),
        stripped=stripped.numpy(),
    )

    # IWAE with a WFST (per-mark) unnormalised model
    torch.manual_seed(5)
    theta_mod = torch.nn.Embedding(V, 1)
    with torch.no_grad():
        theta_mod.weight.copy_(torch.randn(V, 1) * 0.5 - 1.0)
    wfst = WFSTScorer(PAD, BOS, EOS, theta_mod)
    torch.manual_seed(4321)
    with torch.no_grad():
        log_marg, log_q2, samples2, log_w = Estimators.iwae(sampler, wfst, B, K, 0, None)
        seqs = torch.from_numpy(np.random.default_rng(3).integers(0, V, size=(6, 11)))
        wscore = wfst.wfst_score(seqs)
    save(
        "iwae.npz",
        emission=em,
        transition=tr,
        K=np.int64(K),
        theta=theta_mod.weight.detach().numpy().reshape(-1).astype(np.float32),
        log_marginal=log_marg.numpy().astype(np.float32),
        log_q=log_q2.numpy().astype(np.float32),
        samples=samples2.numpy(),
        log_w=log_w.numpy().astype(np.float32),
        wfst_seqs=seqs.numpy(),
        wfst_score=wscore.numpy().astype(np.float32),
    )


class _GatherProbe(StaticRNNScorer):
    """evaluate_seq_with_temp on supplied scores: the RNN (get_seqs) is replaced
    by a table lookup; everything downstream is the reference's own code."""

    def __init__(self, V, max_length, locally_normalized, label_smoothing):
        self.bidirectional = False
        CompositeScorer.__init__(
            self, 4, V, activation=None, pad=PAD, bos=BOS, eos=EOS, max_length=max_length, num_hidden_states=1
        )
        self.simple_sum = False
        self.right_to_left = False
        self.locally_normalized = locally_normalized
        self.label_smoothing = label_smoothing
        self.bptt_warning = True
        self._scores = None

    def get_seqs(self, sequence, **kwargs):
        return self._scores, None


def make_evalseq():
    V, N, T = 20, 6, 9
    rng = np.random.default_rng(8)
    seqs = np.full((N, T), PAD, dtype=np.int64)
    for n in range(N):
        L = int(rng.integers(2, T - 1))
        seqs[n, :L] = rng.integers(3, V, size=L)
        seqs[n, L] = EOS
    seqs[0, :] = rng.integers(3, V, size=T)  # a row that never ends (hits has_to_end masks)
    scores = rng.normal(0, 1.5, size=(N, T, V)).astype(np.float32)
    out = {"seqs": seqs, "scores": scores}
    for tag, (maxlen, norm, smooth, training, temp) in {
        "norm_eval": (30, True, 0.0, False, 1.0),
        "norm_eval_temp": (30, True, 0.0, False, 0.7),
        "norm_eval_short": (5, True, 0.0, False, 1.0),
        "raw_eval": (30, False, 0.0, False, 1.0),
        "norm_train_smooth": (30, True, 0.1, True, 1.0),
    }.items():
        pr = _GatherProbe(V, maxlen, norm, smooth)
        pr.train(training)
        pr._scores = torch.from_numpy(scores)
        with torch.no_grad():
            val = pr.evaluate_seq_with_temp(torch.from_numpy(seqs), temp=temp)
        out[tag] = val.numpy().astype(np.float32)
        out[tag + "_cfg"] = np.array([maxlen, int(norm), smooth, int(training), temp], dtype=np.float64)
    save("evalseq.npz", **out)


def make_evalseq_grad():
    """Gradients of evaluate_seq_with_temp with respect to the supplied scores (the reference trains
    p~ through it, lightning.py:511-516): d (sum_n g[n] * value[n]) / d scores by torch.autograd on the
    reference's own code.  V = 20 (16-byte streaming kernels) and V = 22 (scalar fallback)."""
    out = {}
    for V in (20, 22):
        N, T = 6, 9
        rng = np.random.default_rng(80 + V)
        seqs = np.full((N, T), PAD, dtype=np.int64)
        for n in range(N):
            L = int(rng.integers(2, 6))  # within max_length = 5 of the "short" case
            seqs[n, :L] = rng.integers(3, V, size=L)
            seqs[n, L] = EOS
        scores = rng.normal(0, 1.5, size=(N, T, V)).astype(np.float32)
        g = rng.normal(0, 1.0, size=N).astype(np.float32)
        out.update({f"v{V}_seqs": seqs, f"v{V}_scores": scores, f"v{V}_g": g})
        for tag, (maxlen, norm, smooth, training, temp) in {
            "norm_eval": (30, True, 0.0, False, 1.0),
            "norm_eval_temp": (30, True, 0.0, False, 0.7),
            "norm_eval_short": (5, True, 0.0, False, 1.0),
            "raw_eval": (30, False, 0.0, False, 1.0),
            "norm_train_smooth": (30, True, 0.1, True, 1.0),
            "norm_train_smooth_temp": (30, True, 0.25, True, 1.3),
            "raw_train_smooth": (30, False, 0.2, True, 0.8),
        }.items():
            pr = _GatherProbe(V, maxlen, norm, smooth)
            pr.train(training)
            sc = torch.from_numpy(scores).clone().requires_grad_(True)
            pr._scores = sc
            val = pr.evaluate_seq_with_temp(torch.from_numpy(seqs), temp=temp)
            (val * torch.from_numpy(g)).sum().backward()
            assert torch.isfinite(val).all() and torch.isfinite(sc.grad).all()
            out[f"v{V}_{tag}"] = val.detach().numpy().astype(np.float32)
            out[f"v{V}_{tag}_grad"] = sc.grad.numpy().astype(np.float32)
            out[f"v{V}_{tag}_cfg"] = np.array([maxlen, int(norm), smooth, int(training), temp], dtype=np.float64)
    save("evalseq_grad.npz", **out)


def make_gpt2():
    """GPT2Wrapper.forward (transformer.py:34-52) on supplied logits: the language model inside the
    wrapper is replaced by a stub that returns the supplied tensor; the pad logit overwrite, the
    log_softmax, the gather of the shifted gold sequence, the pad mask and the sum are the reference's
    own code.  Values and d (sum_n g[n] value[n]) / d logits."""
    from types import SimpleNamespace
    from src.modules.transformer import GPT2Wrapper

    class _Stub(torch.nn.Module):
        def forward(self, inp):
            assert tuple(inp.shape) == tuple(self.logits.shape[:2])
            return SimpleNamespace(logits=self.logits * 1.0)  # not a leaf: the wrapper writes into it

    out = {}
    for V in (20, 22, 300):
        N, T = 5, 8
        rng = np.random.default_rng(90 + V)
        x = np.full((N, T), PAD, dtype=np.int64)
        for n in range(N):
            L = int(rng.integers(2, T))
            x[n, :L] = rng.integers(3, V, size=L)
            x[n, L] = EOS
        x[0, :] = rng.integers(3, V, size=T)  # a full row: its only pad is the appended one
        logits = rng.normal(0, 2.0, size=(N, T + 1, V)).astype(np.float32)
        g = rng.normal(0, 1.0, size=N).astype(np.float32)
        # the constructor only builds the Hugging Face model (and its positional GPT2Config call is
        # rejected by the transformers version installed here): the attributes forward() reads are set
        # by hand, forward() itself is the reference's
        w = GPT2Wrapper.__new__(GPT2Wrapper)
        torch.nn.Module.__init__(w)
        w.vocab_size, w.bos, w.eos, w.pad, w.max_length = V, BOS, EOS, PAD, T + 1
        w.model = _Stub()
        lg = torch.from_numpy(logits).clone().requires_grad_(True)
        w.model.logits = lg
        val = w(torch.from_numpy(x))
        (val * torch.from_numpy(g)).sum().backward()
        out.update({f"v{V}_x": x, f"v{V}_logits": logits, f"v{V}_g": g, f"v{V}_value": val.detach().numpy().astype(np.float32),
                    f"v{V}_grad": lg.grad.numpy().astype(np.float32)})
    save("gpt2.npz", **out)


def make_sampler_beta():
    """Sampler.stateful_sample with the learned proposal FSAGRUScorer(use_beta=True) (samplers.py:182-335,
    scorers.py:577-601 + 630-681): per step the recurrent cell's prefix scores (recorded by a forward hook
    on ``beta_scorer``; the network itself is out of scope), the beta values of compute_beta() that are
    added through the next-state gather of scorers.py:584-590, the insertion penalty
    (insert_threshold > 0, scorers.py:663-669) and the length penalty (0 < length_threshold < length,
    671-677).  Outputs: the samples and their log q.  Pins the order of the steps: the beta gather
    reads the transition row of the state *before* the previous symbol is consumed (584-590 run inside
    super().actual_left_to_right_score, the state advance is 679), the masks use the state after it."""
    V = 24
    lats = [
        synth.layered_lattice(31, n_states=20, avg_degree=3.0, vocab=V, width=3, span=2),
        synth.edit_lattice([10, 11, 12], [13, 14], vocab=V, seed=4),  # uses the marks 3, 4, 5 (insertion-mark = 5)
    ]
    em, tr = synth.collate_dense([l.dense() for l in lats], pad=PAD)
    B, K = len(lats), 6
    out = {"emission": em, "transition": tr, "K": np.int64(K)}
    for tag, kw, temperature in (("a", dict(insert_threshold=1, insert_penalty=0.7, length_threshold=3, length_penalty=0.05), 1.0),
                                 ("b", dict(insert_threshold=2, insert_penalty=1000.0, length_threshold=0, length_penalty=1000.0), 0.8)):
        torch.manual_seed(7)
        sc = FSAGRUScorer(hid_dim=8, vocab_size=V, pad=PAD, bos=BOS, eos=EOS, use_beta=True, max_length=48, **kw)
        sc.eval()
        with torch.no_grad():
            sc.Wh.mul_(0.5)
            sc.W.mul_(0.3)  # keeps beta (probability domain, added raw to the logits) of order one
        rec = []
        hook = sc.beta_scorer.register_forward_hook(lambda m, i, o: rec.append(o.detach().clone().numpy()))
        sampler = Sampler(sc)
        sampler.set_masks(transition=torch.from_numpy(tr), emission=torch.from_numpy(em))
        sampler.set_k(K)
        torch.manual_seed(99)
        with torch.no_grad():
            beta = sc.compute_beta().numpy().astype(np.float32)
            log_q, samples, _ = sampler.stateful_sample(B * K, temperature=temperature)
        hook.remove()
        assert len(rec) == samples.shape[1] + 1
        out.update({f"{tag}_beta": beta, f"{tag}_prefix_scores": np.stack(rec, 0).astype(np.float32),
                    f"{tag}_samples": samples.numpy(), f"{tag}_log_q": log_q.numpy().astype(np.float32),
                    f"{tag}_cfg": np.array([kw["insert_threshold"], kw["insert_penalty"], kw["length_threshold"],
                                            kw["length_penalty"], temperature, 48, Vocab.lookup("insertion-mark")], np.float64)})
    save("sampler_beta.npz", **out)


def make_strip():
    """Sampler.stripping_pad (samplers.py:162-180) with pad != 0: marks equal to 0 are dropped, pad marks
    are kept, and a row that ends in a dropped 0 keeps one 0 after its last mark."""
    from types import SimpleNamespace
    out = {}
    rng = np.random.default_rng(21)
    for pad in (0, 7):
        smp = Sampler.__new__(Sampler)
        smp.model = SimpleNamespace(__pad__=pad)
        seqs = rng.integers(0, 10, size=(12, 14)).astype(np.int64)
        seqs[rng.random(seqs.shape) < 0.25] = 0
        for n in range(12):
            seqs[n, int(rng.integers(5, 12)):] = pad  # pad tail; column 11 onwards is pad everywhere
        out[f"pad{pad}_in"] = seqs
        out[f"pad{pad}_out"] = smp.stripping_pad(torch.from_numpy(seqs)).numpy()
    save("strip.npz", **out)


if __name__ == "__main__":
    only = set(sys.argv[1:])  # e.g. `make_golden.py make_beta_neural`; nothing = all
    for fn in (make_beta, make_beta_neural, make_beta_neural_grad, make_gather, make_sampler_and_iwae, make_evalseq, make_evalseq_grad, make_gpt2,
               make_strip, make_sampler_beta):
        if not only or fn.__name__ in only:
            fn()
