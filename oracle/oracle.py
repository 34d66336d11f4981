"""CPU oracle for the nFST lattice hot path -- TEST INFRASTRUCTURE ONLY.

Importable only from ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; it is the checker, never the product, never
a fallback.  Parity status: **pinned** -- ``tests/test_oracle_golden.py`` checks
every function here against ``tests/golden/*.npz`` produced from the reference's
own code by ``tests/golden/make_golden.py``.

The heavy loops live in ``nfst_oracle.c`` (built by ``oracle/Makefile``); the
pure gathers are numpy.  Reference citations are into /root/reference/src.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libnfst_oracle.so")
_lib = None

ERRORS = {-1: "cycle", -2: "not exactly one sink", -3: "bad index", -4: "inconsistent table", -5: "buffer too small"}


class OracleError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "nfst_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_dense_to_arcs.restype = C.c_int64
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def _chk(rc):
    if rc < 0:
        raise OracleError(ERRORS.get(int(rc), f"error {rc}"))
    return rc


# --------------------------------------------------------------------------- tables
def dense_to_arcs(emission: np.ndarray, transition: np.ndarray):
    """Arc list (src, label, dst, weight|None) of one dense lattice in (state asc,
    label asc) order over states reachable from 0 (scorers.py:995-1035 encoding)."""
    n_rows, V = transition.shape
    transition = np.ascontiguousarray(transition, dtype=np.int64)
    cap = int(n_rows) * int(V)
    src = np.empty(cap, np.int32)
    label = np.empty(cap, np.int32)
    dst = np.empty(cap, np.int32)
    if emission.dtype == np.bool_:
        em = np.ascontiguousarray(emission).view(np.uint8)
        n = lib().orc_dense_to_arcs(_p(em, C.c_uint8), None, _p(transition, C.c_int64), n_rows, V,
                                    _p(src, C.c_int32), _p(label, C.c_int32), _p(dst, C.c_int32), None, C.c_int64(cap))
        w = None
    else:
        em = np.ascontiguousarray(emission, dtype=np.float32)
        w = np.empty(cap, np.float32)
        n = lib().orc_dense_to_arcs(None, _p(em, C.c_float), _p(transition, C.c_int64), n_rows, V,
                                    _p(src, C.c_int32), _p(label, C.c_int32), _p(dst, C.c_int32), _p(w, C.c_float), C.c_int64(cap))
    _chk(n)
    n = int(n)
    return src[:n].copy(), label[:n].copy(), dst[:n].copy(), (None if w is None else w[:n].copy())


def update_fsa_state(transition_k: np.ndarray, updated: np.ndarray, prev_states: np.ndarray) -> np.ndarray:
    """scorers.py:683-690: transition_k[arange, prev_states][arange, updated]."""
    n = np.arange(prev_states.shape[0])
    return transition_k[n, prev_states][n, updated]


def expand_k(t: np.ndarray, k: int) -> np.ndarray:
    """scorers.py:887-918 (set_k): [B,S,V] -> [B*k,S,V], each lattice repeated k times."""
    return np.repeat(t, k, axis=0)


def base_step_mask(inp: np.ndarray, V: int, pad: int, bos: int, eos: int, has_to_end: bool) -> np.ndarray:
    """PaddedScorer.mask_invalid_helper (scorers.py:59-83) as used by
    LeftToRightScorer.mask_out_invalid (314-338): float32 [N,V] of 0 / -inf."""
    ninf = np.float32(-np.inf)
    N = inp.shape[0]
    bos_mask = np.zeros(V, np.float32); bos_mask[bos] = ninf
    pad_mask = np.full(V, ninf, np.float32); pad_mask[pad] = 0
    no_pad_mask = np.zeros(V, np.float32); no_pad_mask[pad] = ninf
    eos_mask = np.full(V, ninf, np.float32); eos_mask[eos] = 0
    ended = (inp == eos) | (inp == pad)
    out = np.where(ended[:, None], pad_mask[None, :], no_pad_mask[None, :]) + bos_mask[None, :]
    if has_to_end:
        out = np.where(ended[:, None], np.zeros((N, V), np.float32), eos_mask[None, :]) + out
    return out.astype(np.float32)


def mask_out_invalid(emission_k: np.ndarray, inp: np.ndarray, state: np.ndarray, length: int,
                     max_length: int, pad: int, bos: int, eos: int) -> np.ndarray:
    """FSAGRUScorer.mask_out_invalid (scorers.py:1037-1054)."""
    V = emission_k.shape[2]
    base = base_step_mask(inp, V, pad, bos, eos, length > max_length)
    rows = emission_k[np.arange(state.shape[0]), state]
    if emission_k.dtype == np.bool_:
        st = np.where(rows, np.float32(0), np.float32(-np.inf))
    else:
        st = rows.astype(np.float32)
    return (base + st).astype(np.float32)


def beta_logits(transition_k: np.ndarray, beta: np.ndarray, state: np.ndarray) -> np.ndarray:
    """scorers.py:584-590: gather(beta, 1, transition_k[arange, state])."""
    rows = transition_k[np.arange(state.shape[0]), state]
    return np.take_along_axis(beta, rows, axis=1)


# --------------------------------------------------------------------------- path sums
def forward_backward(n_rows: int, src, dst, score) -> dict:
    """float64 log alpha / log beta / posteriors / log Z of one lattice
    (compute_beta_per_sample semantics, scorers.py:692-751, arc weight exp(score))."""
    src = np.ascontiguousarray(src, np.int32); dst = np.ascontiguousarray(dst, np.int32)
    score = np.ascontiguousarray(score, np.float64)
    A = src.shape[0]
    la = np.empty(n_rows); lb = np.empty(n_rows); post = np.empty(A); z = C.c_double()
    _chk(lib().orc_forward_backward(n_rows, C.c_int64(A), _p(src, C.c_int32), _p(dst, C.c_int32), _p(score, C.c_double),
                                    _p(la, C.c_double), _p(lb, C.c_double), _p(post, C.c_double), C.byref(z)))
    return {"logalpha": la, "logbeta": lb, "posterior": post, "logZ": z.value}


def forward_backward_batch(n_rows, arc_off, src, label, dst, arc_w, theta, n_threads: int = 1,
                           want_posterior: bool = True) -> Tuple[np.ndarray, Optional[np.ndarray]]:
    n_rows = np.ascontiguousarray(n_rows, np.int32); arc_off = np.ascontiguousarray(arc_off, np.int64)
    src = np.ascontiguousarray(src, np.int32); label = np.ascontiguousarray(label, np.int32)
    dst = np.ascontiguousarray(dst, np.int32); theta = np.ascontiguousarray(theta, np.float32)
    if arc_w is not None:
        arc_w = np.ascontiguousarray(arc_w, np.float32)
    B = n_rows.shape[0]
    logz = np.empty(B)
    post = np.empty(src.shape[0]) if want_posterior else None
    _chk(lib().orc_forward_backward_batch(B, _p(n_rows, C.c_int32), _p(arc_off, C.c_int64), _p(src, C.c_int32),
                                          _p(label, C.c_int32), _p(dst, C.c_int32), _p(arc_w, C.c_float),
                                          _p(theta, C.c_float), int(n_threads), _p(logz, C.c_double), _p(post, C.c_double)))
    return logz, post


def viterbi(n_rows: int, src, label, dst, score_f32, max_len: int):
    src = np.ascontiguousarray(src, np.int32); label = np.ascontiguousarray(label, np.int32)
    dst = np.ascontiguousarray(dst, np.int32); score = np.ascontiguousarray(score_f32, np.float32)
    best = C.c_float(); n = C.c_int32()
    path = np.empty(max_len, np.int32); arcs = np.empty(max_len, np.int32)
    _chk(lib().orc_viterbi(n_rows, C.c_int64(src.shape[0]), _p(src, C.c_int32), _p(label, C.c_int32), _p(dst, C.c_int32),
                           _p(score, C.c_float), C.byref(best), _p(path, C.c_int32), _p(arcs, C.c_int32), max_len, C.byref(n)))
    return float(best.value), path[: n.value].copy(), arcs[: n.value].copy()


def sample_paths(n_rows: int, src, label, dst, score, logbeta, uniforms: np.ndarray, pad: int):
    src = np.ascontiguousarray(src, np.int32); label = np.ascontiguousarray(label, np.int32)
    dst = np.ascontiguousarray(dst, np.int32); score = np.ascontiguousarray(score, np.float64)
    logbeta = np.ascontiguousarray(logbeta, np.float64); uniforms = np.ascontiguousarray(uniforms, np.float64)
    K, T = uniforms.shape
    paths = np.empty((K, T), np.int32); arcs = np.empty((K, T), np.int32); lens = np.empty(K, np.int32)
    logq = np.empty(K); margin = np.empty(K)
    _chk(lib().orc_sample_paths(n_rows, C.c_int64(src.shape[0]), _p(src, C.c_int32), _p(label, C.c_int32), _p(dst, C.c_int32),
                                _p(score, C.c_double), _p(logbeta, C.c_double), K, T, _p(uniforms, C.c_double), pad,
                                _p(paths, C.c_int32), _p(arcs, C.c_int32), _p(lens, C.c_int32), _p(logq, C.c_double), _p(margin, C.c_double)))
    return {"paths": paths, "arcs": arcs, "lengths": lens, "logq": logq, "margin": margin}


def score_paths(n_rows: int, src, label, dst, score, marks: np.ndarray):
    src = np.ascontiguousarray(src, np.int32); label = np.ascontiguousarray(label, np.int32)
    dst = np.ascontiguousarray(dst, np.int32); score = np.ascontiguousarray(score, np.float64)
    marks = np.ascontiguousarray(marks, np.int32)
    K, T = marks.shape
    tot = np.empty(K); end = np.empty(K, np.int32)
    _chk(lib().orc_score_paths(n_rows, C.c_int64(src.shape[0]), _p(src, C.c_int32), _p(label, C.c_int32), _p(dst, C.c_int32),
                               _p(score, C.c_double), K, T, _p(marks, C.c_int32), _p(tot, C.c_double), _p(end, C.c_int32)))
    return tot, end


def beta_dense_frontier(transition: np.ndarray, emb, Wx, Wh, W, bias, n_threads: int = 1) -> Tuple[np.ndarray, int]:
    """compute_beta_parallel (scorers.py:753-856) restated; float32 probability domain.  ``n_threads``
    OpenMP threads share the rows of the dense [S, S] cell table (results do not depend on it)."""
    transition = np.ascontiguousarray(transition, np.int64)
    S, V = transition.shape
    emb = np.ascontiguousarray(emb, np.float32); H = emb.shape[1]
    Wx = np.ascontiguousarray(Wx, np.float32); Wh = np.ascontiguousarray(Wh, np.float32)
    W = np.ascontiguousarray(W, np.float32).reshape(-1); bias = np.ascontiguousarray(bias, np.float32)
    if H > 512:
        raise ValueError("H <= 512")
    beta = np.empty(S, np.float32)
    it = lib().orc_beta_dense_frontier(_p(transition, C.c_int64), S, V, H, _p(emb, C.c_float), _p(Wx, C.c_float),
                                       _p(Wh, C.c_float), _p(W, C.c_float), _p(bias, C.c_float), _p(beta, C.c_float),
                                       int(n_threads))
    return beta, int(it)


def beta_neural(n_rows: int, src, label, dst, emb, Wx, Wh, W, bias, arc_w=None):
    """compute_beta_per_sample with its Tree-LSTM-style messages (scorers.py:692-751), restated in
    float64 with beta kept in log space: for every arc (s, l, s') with s' != s
        t = tanh(Wx e(l) + Wh beta_hat(s') + bias),   msg = exp(W . t) * beta(s')       (:732-738)
        beta(s) = sum msg,   beta_hat(s) = sum (msg / beta(s)) t                         (:743-747)
    from beta(sink) = 1, beta_hat(sink) = 0 (:699-701, 720).  Parallel arcs count once each, as in
    the per-sample loop.  ``arc_w`` (float emission tables, scorers.py:1011-1027; not used by the
    reference's beta) adds a log weight per arc to W . t.  Returns (log beta [n_rows], beta_hat
    [n_rows, H]); states that do not reach the sink get -inf / 0."""
    src = np.asarray(src, np.int64); label = np.asarray(label, np.int64); dst = np.asarray(dst, np.int64)
    emb = np.asarray(emb, np.float64); Wx = np.asarray(Wx, np.float64); Wh = np.asarray(Wh, np.float64)
    W = np.asarray(W, np.float64).reshape(-1); bias = np.asarray(bias, np.float64)
    H = emb.shape[1]
    keep = src != dst
    aw = np.zeros(src.shape[0]) if arc_w is None else np.asarray(arc_w, np.float64)
    src, label, dst, aw = src[keep], label[keep], dst[keep], aw[keep]
    x = emb @ Wx.T + bias  # [V, H]
    out = [[] for _ in range(n_rows)]
    pending = np.zeros(n_rows, np.int64)
    into = [[] for _ in range(n_rows)]
    for a in range(src.shape[0]):
        out[src[a]].append(a)
        into[dst[a]].append(a)
        pending[src[a]] += 1
    sinks = [s for s in range(n_rows) if pending[s] == 0 and into[s]]
    if len(sinks) != 1:
        raise OracleError("not exactly one sink")
    logb = np.full(n_rows, -np.inf)
    bhat = np.zeros((n_rows, H))
    logb[sinks[0]] = 0.0
    ready = [sinks[0]]
    while ready:
        s2 = ready.pop()
        for a in into[s2]:
            s = src[a]
            pending[s] -= 1
            if pending[s] == 0:
                arcs = out[s]
                t = np.tanh(x[label[arcs]] + bhat[dst[arcs]] @ Wh.T)  # [n, H]
                lm = t @ W + aw[arcs] + logb[dst[arcs]]
                mx = lm.max()
                if np.isneginf(mx):
                    continue
                logb[s] = mx + np.log(np.exp(lm - mx).sum())
                bhat[s] = np.exp(lm - logb[s]) @ t
                ready.append(s)
    return logb, bhat


def beta_neural_grad(n_rows: int, src, label, dst, emb, Wx, Wh, W, bias, coef, arc_w=None, coef_hat=None):
    """Gradients of L = sum_s coef[s] log beta(s) (+ sum coef_hat[s] . beta_hat(s)) through the
    recurrence of ``beta_neural`` (scorers.py:692-751) with respect to emb, Wx, Wh, W, bias:
    torch.autograd in float64 over an out-of-place restatement (the reference's own functions update
    tensors in place and autograd refuses them; the forward values of this restatement are pinned by
    beta_neural.npz, its derivatives by the central differences of the reference's forward pass in
    beta_neural_grad.npz).  States with coef != 0 must reach the sink.  Returns (loss, log_beta,
    beta_hat, dict of gradients) as numpy float64."""
    import torch
    f64 = torch.float64
    src = np.asarray(src, np.int64); label = np.asarray(label, np.int64); dst = np.asarray(dst, np.int64)
    keep = src != dst
    aw = np.zeros(src.shape[0]) if arc_w is None else np.asarray(arc_w, np.float64)
    src, label, dst, aw = src[keep], label[keep], dst[keep], aw[keep]
    P = {k: torch.tensor(np.asarray(v, np.float64), dtype=f64, requires_grad=True)
         for k, v in dict(emb=emb, Wx=Wx, Wh=Wh, W=np.asarray(W).reshape(-1), bias=bias).items()}
    H = P["emb"].shape[1]
    x = P["emb"] @ P["Wx"].T + P["bias"]
    out = [[] for _ in range(n_rows)]
    into = [[] for _ in range(n_rows)]
    pending = np.zeros(n_rows, np.int64)
    for a in range(src.shape[0]):
        out[src[a]].append(a); into[dst[a]].append(a); pending[src[a]] += 1
    sinks = [s for s in range(n_rows) if pending[s] == 0 and into[s]]
    if len(sinks) != 1:
        raise OracleError("not exactly one sink")
    logb = [None] * n_rows
    bhat = [None] * n_rows
    logb[sinks[0]] = torch.zeros((), dtype=f64)
    bhat[sinks[0]] = torch.zeros(H, dtype=f64)
    ready = [sinks[0]]
    aw_t = torch.tensor(aw, dtype=f64)
    while ready:
        s2 = ready.pop()
        for a in into[s2]:
            s = src[a]
            pending[s] -= 1
            if pending[s] == 0:
                arcs = out[s]
                bh = torch.stack([bhat[dst[i]] for i in arcs])
                lb = torch.stack([logb[dst[i]] for i in arcs])
                t = torch.tanh(x[torch.as_tensor(label[arcs])] + bh @ P["Wh"].T)
                lm = t @ P["W"] + aw_t[torch.as_tensor(arcs)] + lb
                logb[s] = torch.logsumexp(lm, 0)
                bhat[s] = torch.softmax(lm, 0) @ t
                ready.append(s)
    coef = np.asarray(coef, np.float64)
    loss = torch.zeros((), dtype=f64)
    for s in range(n_rows):
        if coef[s] != 0.0:
            loss = loss + coef[s] * logb[s]
        if coef_hat is not None and logb[s] is not None:
            loss = loss + (torch.tensor(np.asarray(coef_hat[s], np.float64)) * bhat[s]).sum()
    loss.backward()
    lb_np = np.array([float(v.detach()) if v is not None else -np.inf for v in logb])
    bh_np = np.stack([v.detach().numpy() if v is not None else np.zeros(H) for v in bhat])
    grads = {k: (v.grad.numpy() if v.grad is not None else np.zeros(tuple(v.shape))) for k, v in P.items()}
    return float(loss.detach()), lb_np, bh_np, grads


# --------------------------------------------------------------------------- estimator side
def stripping_pad(seqs: np.ndarray, pad: int) -> np.ndarray:
    """Sampler.stripping_pad (samplers.py:162-180): drop marks equal to 0, left-align."""
    N, T = seqs.shape
    out = np.full_like(seqs, pad)
    idx = np.zeros(N, np.int64)
    i = 0
    for i in range(T):
        out[np.arange(N), idx] = seqs[:, i]
        idx = idx + (seqs[:, i] != 0)
        if np.all(seqs[:, i] == pad):
            break
    return out[:, : i + 1].copy()


def wfst_score(theta: np.ndarray, t: np.ndarray, pad: int) -> np.ndarray:
    """WFSTScorer.wfst_score (scorers.py:1671-1687): sum of theta[mark] over non-pad marks."""
    sc = theta.astype(np.float32)[t]
    return np.where(t == pad, np.float32(0), sc).sum(axis=1, dtype=np.float32)


def iwae(log_p: np.ndarray, log_q: np.ndarray):
    """Estimators.iwae (estimatros.py:11-44): log_w = log p - log q; logsumexp_k - log k."""
    log_w = (log_p - log_q).astype(np.float32)
    k = log_w.shape[1]
    m = log_w.max(axis=1, keepdims=True)
    lse = (m[:, 0] + np.log(np.exp(log_w - m).sum(axis=1))).astype(np.float32)
    return (lse - np.float32(np.log(k))).astype(np.float32), log_w


def batched_seq_mask(seqs: np.ndarray, V: int, pad: int, bos: int, eos: int, max_length: Optional[int]) -> np.ndarray:
    """PaddedScorer.slow_but_correct_batched_mask_out_invalid (scorers.py:89-134)."""
    N, T = seqs.shape
    out = np.zeros((N, T, V), np.float32)
    out[:, 0, bos] = -np.inf
    out[:, 0, pad] = -np.inf
    for idx in range(T - 1):
        flag = False if max_length is None else (idx + 1 > max_length)
        out[:, idx + 1] = base_step_mask(seqs[:, idx], V, pad, bos, eos, flag)
    return out


def evaluate_seq(scores: np.ndarray, seqs: np.ndarray, pad: int, bos: int, eos: int, max_length: int,
                 temp: float = 1.0, normalize: bool = True, training: bool = False, smoothing: float = 0.0) -> np.ndarray:
    """StaticRNNScorer.evaluate_seq_with_temp arithmetic (scorers.py:1564-1611) on
    supplied scores [N,T,V]: masks, log_softmax, gather (or label-smoothed dot
    product when training, 1502-1528 + 1584-1592), pad masking, sum over time."""
    N, T, V = scores.shape
    mask = batched_seq_mask(seqs, V, pad, bos, eos, max_length).astype(np.float64)
    pz = np.ones(V); pz[pad] = 0
    with np.errstate(invalid="ignore", over="ignore"):
        through = (scores.astype(np.float64) * pz[None, None, :] + mask) / temp
        if normalize:
            x = through + mask
            m = x.max(axis=2, keepdims=True)
            final = x - (m + np.log(np.exp(x - m).sum(axis=2, keepdims=True)))
        else:
            final = through + mask
        flat = final.reshape(-1, V)
        lab = seqs.reshape(-1)
        if training:
            finite = flat > -np.inf
            cnt = finite.sum(axis=1, keepdims=True).astype(np.float64)
            dist = np.where(finite, smoothing / (cnt - 1), 0.0) * np.ones_like(flat)
            dist[np.arange(flat.shape[0]), lab] = 1.0 - smoothing
            sel = (dist * np.clip(flat, -10e8, 10e8)).sum(axis=1)
        else:
            sel = flat[np.arange(flat.shape[0]), lab]
        sel = sel.reshape(N, T) * (seqs != pad)
    return sel.sum(axis=1)


def evaluate_seq_grad(scores: np.ndarray, seqs: np.ndarray, g: np.ndarray, pad: int, bos: int, eos: int, max_length: int,
                      temp: float = 1.0, normalize: bool = True, training: bool = False, smoothing: float = 0.0) -> np.ndarray:
    """d (sum_n g[n] evaluate_seq(...)[n]) / d scores, float64 -- what torch.autograd gives through
    scorers.py:1564-1611: with e = (scores * padmask + mask) / temp + mask, final = log_softmax(e) or e,
    value = sum_v td[v] clamp(final[v]) per row (td: one-hot, or the smoothed target of 1502-1528),
        d value / d scores[u] = (td[u] c'[u] - softmax(e)[u] * sum_v td[v] c'[v]) * padmask[u] / temp
    (the softmax term only with the log_softmax; c' = the clamp's derivative)."""
    N, T, V = scores.shape
    mask = batched_seq_mask(seqs, V, pad, bos, eos, max_length).astype(np.float64)
    pz = np.ones(V); pz[pad] = 0
    with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
        e = (scores.astype(np.float64) * pz[None, None, :] + mask) / temp + mask
        if normalize:
            m = e.max(axis=2, keepdims=True)
            lse = m + np.log(np.exp(e - m).sum(axis=2, keepdims=True))
            final = e - lse
            p = np.exp(final)
        else:
            final = e
            p = np.zeros_like(e)
        lab = seqs.reshape(-1)
        flat = final.reshape(-1, V)
        rows = np.arange(flat.shape[0])
        if training:
            finite = flat > -np.inf
            cnt = finite.sum(axis=1, keepdims=True).astype(np.float64)
            td = np.where(finite, smoothing / (cnt - 1), 0.0) * np.ones_like(flat)
            td[rows, lab] = 1.0 - smoothing
            td = td * ((flat >= -10e8) & (flat <= 10e8))
        else:
            td = np.zeros_like(flat)
            td[rows, lab] = 1.0
        W = td.sum(axis=1, keepdims=True) if normalize else 0.0
        d = (td - p.reshape(-1, V) * W).reshape(N, T, V)
        d = d * pz[None, None, :] / temp
        d = d * ((seqs != pad)[:, :, None]) * g.astype(np.float64)[:, None, None]
    return d


def gpt2_logprob(logits: np.ndarray, x: np.ndarray, pad: int, g: Optional[np.ndarray] = None):
    """GPT2Wrapper.forward after the language model (transformer.py:38-52): gold = cat(x, pad), the pad
    logit becomes -1e8, log_softmax, gather, positions holding pad contribute 0, sum over time.
    With ``g`` also d (sum_n g[n] value[n]) / d logits (the overwritten pad column gets none)."""
    N, T1, V = logits.shape
    gold = np.concatenate([x, np.full((N, 1), pad, x.dtype)], axis=1)
    lg = logits.astype(np.float64).copy()
    lg[:, :, pad] = -1e8
    m = lg.max(axis=2, keepdims=True)
    lp = lg - (m + np.log(np.exp(lg - m).sum(axis=2, keepdims=True)))
    keep = gold != pad
    val = (np.take_along_axis(lp, gold[:, :, None], axis=2)[:, :, 0] * keep).sum(axis=1)
    if g is None:
        return val
    onehot = np.zeros_like(lp)
    np.put_along_axis(onehot, gold[:, :, None], 1.0, axis=2)
    d = (onehot - np.exp(lp)) * keep[:, :, None] * g.astype(np.float64)[:, None, None]
    d[:, :, pad] = 0.0
    return val, d


def proposal_step(emission_k: np.ndarray, transition_k: np.ndarray, scores: np.ndarray, inp: np.ndarray,
                  state: np.ndarray, length: int, max_length: int, pad: int, bos: int, eos: int,
                  temperature: float = 1.0, beta: Optional[np.ndarray] = None, uniforms: Optional[np.ndarray] = None,
                  forced: Optional[np.ndarray] = None, value_state: Optional[np.ndarray] = None,
                  penalties: Optional[dict] = None) -> dict:
    """One step of Sampler.stateful_sample (samplers.py:243-297) on the lattice side:
    left_to_right_score (scorers.py:340-366: pad_masking(scores) + mask_out_invalid, identity
    activation; with beta the gathered beta "logits" of scorers.py:581-593 are added to the scores),
    / temperature, Categorical(logits) -> sample (inverse CDF over the marks in id order on
    ``uniforms``) or evaluate ``forced``, log_prob, logsumexp, update_fsa_state (scorers.py:683-690).
    ``state`` is the state after the previous symbol ``inp`` was consumed (the masks' state);
    ``value_state`` the state the beta gather reads its transition row from -- in the reference the
    state *before* ``inp`` was consumed (scorers.py:584-590 run before the advance at 679); None = ``state``.
    ``penalties``: dict(accumulated [N] int, vocab_use [N, V] float -- both updated in place with
    ``inp`` as scorers.py:654-661 does --, insertion_mark, insert_threshold, insert_penalty,
    length_threshold, length_penalty): the insertion and length penalties of scorers.py:663-677.
    float64 arithmetic; ``margin`` = distance of u from the nearest CDF boundary."""
    N, V = scores.shape
    x = scores.astype(np.float64).copy()
    if beta is not None:
        x = x + beta_logits(transition_k, beta, state if value_state is None else value_state).astype(np.float64)
    if penalties is not None:
        pn = penalties
        pn["accumulated"] += (inp == pn["insertion_mark"])
        pn["vocab_use"][np.arange(N), inp] += 1
        if pn["insert_threshold"] > 0:
            hit = pn["accumulated"] > pn["insert_threshold"]
            x[hit, pn["insertion_mark"]] -= pn["insert_penalty"] * (length - pn["insert_threshold"])
        if 0 < pn["length_threshold"] < length:
            x = x - pn["vocab_use"].astype(np.float64) * pn["length_penalty"]
    x[:, pad] = 0.0  # pad_masking of the summed scores: the pad column counts as 0 (scorers.py:182-187, 357)
    x = (x + mask_out_invalid(emission_k, inp, state, length, max_length, pad, bos, eos).astype(np.float64)) / temperature
    mx = x.max(axis=1, keepdims=True)
    with np.errstate(invalid="ignore", divide="ignore"):
        logz = (mx + np.log(np.exp(x - mx).sum(axis=1, keepdims=True)))[:, 0]
        p = np.exp(x - logz[:, None])
    p = np.where(np.isfinite(x), p, 0.0)
    cdf = np.cumsum(p, axis=1)
    margin = np.full(N, np.inf)
    if forced is not None:
        sym = forced.astype(np.int64)
    else:
        sym = np.empty(N, np.int64)
        for n in range(N):
            legal = np.nonzero(p[n] > 0)[0]
            if legal.size == 0:  # no legal mark: the reference's Categorical would raise
                sym[n], margin[n] = pad, 0.0
                continue
            hit = legal[uniforms[n] < cdf[n, legal]]
            sym[n] = hit[0] if hit.size else legal[-1]
            margin[n] = np.min(np.abs(cdf[n, legal] - uniforms[n]))
    logq = x[np.arange(N), sym] - logz
    nxt = update_fsa_state(transition_k, sym, state)
    return {"symbol": sym, "logq": logq, "logz": logz, "next_state": nxt, "margin": margin}
