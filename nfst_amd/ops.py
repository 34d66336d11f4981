"""Operators of the lattice engine: thin wrappers that allocate outputs with torch
and launch the HIP kernels through the C ABI (include/nfst_hip.h) on torch's
current stream.  There is no CPU implementation: every function raises if the
batch is not on a HIP device.  Reference call sites replaced are cited per op.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Optional

import torch

from . import _lib
from ._lib import lib, check
from .lattice import LatticeBatch


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(lat: LatticeBatch) -> None:
    if lat.device.type != "cuda":
        raise RuntimeError("nfst_amd: the lattice engine runs on the MI355X only (no CPU fallback); "
                           "move the batch with LatticeBatch.to('cuda')")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _scores(lat: LatticeBatch, theta: torch.Tensor, arc_scores: Optional[torch.Tensor]):
    """nfst_scores + the tensors that must stay alive during the launch."""
    if theta.dtype != torch.float32 or theta.device != lat.device:
        theta = theta.to(device=lat.device, dtype=torch.float32)
    theta = theta.contiguous()
    if theta.dim() == 1:
        if theta.shape[0] != lat.vocab:
            raise ValueError(f"theta must have {lat.vocab} entries")
        stride = 0
    elif theta.dim() == 2 and theta.shape == (lat.n_lattices, lat.vocab):
        stride = lat.vocab
    else:
        raise ValueError("theta must be [V] or [B, V]")
    if arc_scores is not None:
        arc_scores = arc_scores.to(device=lat.device, dtype=torch.float32).contiguous()
        if arc_scores.shape != (lat.total_arcs,):
            raise ValueError(f"arc_scores must be [{lat.total_arcs}] in canonical arc order")
        if arc_scores.data_ptr() % 16:
            # a slice or split of a larger score tensor is contiguous but starts anywhere; the kernels read
            # per-arc scores 16 bytes at a time (nfst_forward_backward refuses a misaligned base)
            arc_scores = arc_scores.clone()
    return _lib.Scores(theta.data_ptr(), stride, _ptr(arc_scores), None, 0), (theta, arc_scores)


class BackwardResult(NamedTuple):
    logbeta: Optional[torch.Tensor]  # [total_rows] float32
    logz: torch.Tensor  # [B] float32
    logz64: torch.Tensor  # [B] float64
    beta_me: Optional[torch.Tensor]  # [total_rows, 2] float32 (mantissa, exponent bits)


def backward(lat: LatticeBatch, theta, arc_scores=None, want_logbeta=True, want_me=False) -> BackwardResult:
    """beta sweep + log Z.  Replaces FSAGRUScorer.compute_beta
    (/root/reference/src/modules/scorers.py:858-875)."""
    _need_gpu(lat)
    sc, keep = _scores(lat, theta, arc_scores)
    dev = lat.device
    logbeta = torch.empty(lat.total_rows, dtype=torch.float32, device=dev) if want_logbeta else None
    z64 = torch.empty(lat.n_lattices, dtype=torch.float64, device=dev)
    z32 = torch.empty(lat.n_lattices, dtype=torch.float32, device=dev)
    me = torch.empty((lat.total_rows, 2), dtype=torch.float32, device=dev) if want_me else None
    check(lib.nfst_backward(C.byref(lat.c_struct()), C.byref(sc), _ptr(logbeta), _ptr(z64), _ptr(z32), _ptr(me),
                            _stream()), "nfst_backward")
    return BackwardResult(logbeta, z32, z64, me)


class ForwardBackwardResult(NamedTuple):
    logz: torch.Tensor
    logz64: torch.Tensor
    logalpha: Optional[torch.Tensor]
    logbeta: Optional[torch.Tensor]
    posterior: Optional[torch.Tensor]  # [total_arcs] canonical order
    grad_theta: Optional[torch.Tensor]  # [B, V]
    beta_me: Optional[torch.Tensor]


def forward_backward(lat: LatticeBatch, theta, arc_scores=None, want_alpha_beta=True, want_posterior=True,
                     want_grad_theta=False, want_me=False,
                     out: Optional[ForwardBackwardResult] = None, total=None, total_slot: int = 0) -> ForwardBackwardResult:
    """alpha/beta sweeps, exact log Z and arc posteriors (the quantity the
    reference only estimates by IWAE, modules/estimatros.py:33-44).  ``out`` (a
    previous result of the same batch and flags) is overwritten in place instead of
    allocating new outputs -- the steady state of a training loop.  ``total`` (a float64 tensor
    of 3 zeros, owned by the caller) receives sum_b log Z[b] in ``total[total_slot]`` without a
    reduction kernel; pass ``total_slot = step % 3`` (the launch clears the next slot)."""
    _need_gpu(lat)
    sc, keep = _scores(lat, theta, arc_scores)
    dev = lat.device
    f32 = dict(dtype=torch.float32, device=dev)
    if out is not None:
        z32, z64, la, lb, post, gth, me = out
        if ((la is None) == want_alpha_beta or (post is None) == want_posterior or (gth is None) == want_grad_theta
                or (me is None) == want_me or z64.shape[0] != lat.n_lattices):
            raise ValueError("`out` was produced with different flags or for another batch")
    else:
        la = torch.empty(lat.total_rows, **f32) if want_alpha_beta else None
        lb = torch.empty(lat.total_rows, **f32) if want_alpha_beta else None
        z64 = torch.empty(lat.n_lattices, dtype=torch.float64, device=dev)
        z32 = torch.empty(lat.n_lattices, **f32)
        post = torch.empty(lat.total_arcs, **f32) if want_posterior else None
        gth = torch.empty((lat.n_lattices, lat.vocab), **f32) if want_grad_theta else None
        me = torch.empty((lat.total_rows, 2), **f32) if want_me else None
    if total is not None and (total.dtype != torch.float64 or total.numel() != 3 or total.device != z64.device):
        raise ValueError("`total` must be a float64 tensor of 3 elements on the batch's device")
    check(lib.nfst_forward_backward(C.byref(lat.c_struct()), C.byref(sc), _ptr(la), _ptr(lb), _ptr(z64), _ptr(z32),
                                    _ptr(post), _ptr(gth), _ptr(me), _ptr(total), int(total_slot), _stream()),
          "nfst_forward_backward")
    return ForwardBackwardResult(z32, z64, la, lb, post, gth, me)


class ForwardBackwardLaunch:
    """``forward_backward`` with everything a call computes on the host computed ONCE: the steady state of a training
    loop -- same batch, same score tensors (updated in place by the optimiser), outputs overwritten -- pays one ctypes call
    per step (~5 us) instead of the wrapper's argument checks and allocations (~40 us: longer than the 38 us the kernel takes
    on the BASELINE batch, so the GPU idled between launches).  ``launch = ForwardBackwardLaunch(lat, theta, ...)``;
    ``launch(total_slot)`` enqueues the step on torch's current stream and returns ``launch.out`` (a
    ``ForwardBackwardResult`` whose tensors are overwritten by every launch).  The tensors passed in are held; replacing
    them (instead of updating them in place) needs a new object."""

    def __init__(self, lat: LatticeBatch, theta, arc_scores=None, want_alpha_beta=True, want_posterior=True, want_grad_theta=False,
                 want_me=False, out: Optional[ForwardBackwardResult] = None, total=None):
        _need_gpu(lat)
        self.lat = lat
        self._sc, self._keep = _scores(lat, theta, arc_scores)
        self.out = forward_backward(lat, theta, arc_scores=arc_scores, want_alpha_beta=want_alpha_beta, want_posterior=want_posterior,
                                    want_grad_theta=want_grad_theta, want_me=want_me, out=out)  # (allocates / checks everything, once)
        if total is not None and (total.dtype != torch.float64 or total.numel() != 3 or total.device != lat.device):
            raise ValueError("`total` must be a float64 tensor of 3 elements on the batch's device")
        self.total = total
        z32, z64, la, lb, post, gth, me = self.out
        self._struct = lat.c_struct()
        vp = C.c_void_p
        self._args = (C.byref(self._struct), C.byref(self._sc), vp(_ptr(la)), vp(_ptr(lb)), vp(_ptr(z64)), vp(_ptr(z32)), vp(_ptr(post)),
                      vp(_ptr(gth)), vp(_ptr(me)), vp(_ptr(total)))
        self._fn = lib.nfst_forward_backward

    def __call__(self, total_slot: int = 0) -> "ForwardBackwardResult":
        rc = self._fn(*self._args, total_slot, torch.cuda.current_stream().cuda_stream)
        if rc:
            check(rc, "nfst_forward_backward")
        return self.out


class _LogZ(torch.autograd.Function):
    """log Z with d log Z / d score = arc posterior."""

    @staticmethod
    def forward(ctx, lat, theta, arc_scores):
        need_t = theta.requires_grad
        need_a = arc_scores is not None and arc_scores.requires_grad
        r = forward_backward(lat, theta.detach(), None if arc_scores is None else arc_scores.detach(),
                             want_alpha_beta=False, want_posterior=need_a, want_grad_theta=need_t)
        ctx.lat = lat
        ctx.shared_theta = theta.dim() == 1
        ctx.save_for_backward(r.posterior if need_a else None, r.grad_theta if need_t else None)
        return r.logz

    @staticmethod
    def backward(ctx, g):
        post, gth = ctx.saved_tensors
        g_theta = g_arc = None
        if gth is not None:
            g_theta = gth * g[:, None]
            if ctx.shared_theta:
                g_theta = g_theta.sum(dim=0)
        if post is not None:
            g_arc = post * g[ctx.lat.arc_lattice()]
        return None, g_theta, g_arc


def log_z(lat: LatticeBatch, theta: torch.Tensor, arc_scores: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Differentiable exact log-marginal per lattice, float32 [B]."""
    return _LogZ.apply(lat, theta, arc_scores)


class ViterbiResult(NamedTuple):
    best: torch.Tensor  # [B] float32
    paths: torch.Tensor  # [B, max_len] int32, pad-terminated (bos .. eos)
    arcs: torch.Tensor  # [B, max_len] int32 canonical arc ids, -1 padded
    lengths: torch.Tensor  # [B] int32


def viterbi(lat: LatticeBatch, theta, arc_scores=None, max_len: Optional[int] = None, pad: int = 0) -> ViterbiResult:
    """Best path per lattice (best_sample of JointProb.forward, modules/lightning.py:474-479)."""
    _need_gpu(lat)
    sc, keep = _scores(lat, theta, arc_scores)
    if max_len is None:
        max_len = int(lat.depth.max()) + 1
    dev = lat.device
    best = torch.empty(lat.n_lattices, dtype=torch.float32, device=dev)
    paths = torch.empty((lat.n_lattices, max_len), dtype=torch.int32, device=dev)
    arcs = torch.empty((lat.n_lattices, max_len), dtype=torch.int32, device=dev)
    lens = torch.empty(lat.n_lattices, dtype=torch.int32, device=dev)
    check(lib.nfst_viterbi(C.byref(lat.c_struct()), C.byref(sc), _ptr(best), _ptr(paths), _ptr(arcs), _ptr(lens),
                           int(max_len), int(pad), _stream()), "nfst_viterbi")
    return ViterbiResult(best, paths, arcs, lens)


class SampleResult(NamedTuple):
    paths: torch.Tensor  # [B, K, max_len] int32 labels, pad-terminated
    arcs: torch.Tensor  # [B, K, max_len] int32 canonical arc ids
    lengths: torch.Tensor  # [B, K]
    logq: torch.Tensor  # [B, K] float32 = path score - log Z
    logz: torch.Tensor  # [B] float32


def sample_paths(lat: LatticeBatch, theta, k: int, arc_scores=None, max_len: Optional[int] = None,
                 uniforms: Optional[torch.Tensor] = None, seed: int = 0, pad: int = 0,
                 beta: Optional[BackwardResult] = None, want_arcs: bool = True) -> SampleResult:
    """K exact posterior samples per lattice (Sampler.sample, modules/samplers.py:137-335,
    with the exact posterior as proposal).  ``want_arcs=False`` leaves ``arcs`` None (the C entry
    point's path_arcs is optional; with it the kernel precomputes every arc's probability once per
    block, without it a walk computes the probabilities of the arcs it meets)."""
    _need_gpu(lat)
    sc, keep = _scores(lat, theta, arc_scores)
    if max_len is None:
        max_len = int(lat.depth.max()) + 1
    if beta is None or beta.beta_me is None:
        beta = backward(lat, theta, arc_scores, want_logbeta=False, want_me=True)
    dev = lat.device
    B = lat.n_lattices
    if uniforms is not None:
        uniforms = uniforms.to(device=dev, dtype=torch.float32).contiguous()
        if uniforms.shape != (B, k, max_len):
            raise ValueError(f"uniforms must be [{B}, {k}, {max_len}]")
    paths = torch.empty((B, k, max_len), dtype=torch.int32, device=dev)
    arcs = torch.empty((B, k, max_len), dtype=torch.int32, device=dev) if want_arcs else None
    lens = torch.empty((B, k), dtype=torch.int32, device=dev)
    logq = torch.empty((B, k), dtype=torch.float32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    check(lib.nfst_sample_paths(C.byref(lat.c_struct()), C.byref(sc), _ptr(beta.beta_me), _ptr(beta.logz64), int(k),
                                int(max_len), _ptr(uniforms), C.c_uint64(seed & (2 ** 64 - 1)), int(pad), _ptr(paths),
                                _ptr(arcs), _ptr(lens), _ptr(logq), _ptr(status), _stream()), "nfst_sample_paths")
    st = int(status.item())
    if st != 0:
        raise _lib.NfstError(st, "nfst_sample_paths")  # "ran out of length budget" (samplers.py:299-302)
    return SampleResult(paths, arcs, lens, logq, beta.logz)


def score_paths(lat: LatticeBatch, theta, marks: torch.Tensor, arc_scores=None):
    """Forced walk of marks [B, K, T] (samplers.py:208-218): (path_score [B,K], end_state [B,K])."""
    _need_gpu(lat)
    sc, keep = _scores(lat, theta, arc_scores)
    marks = marks.to(device=lat.device, dtype=torch.int32).contiguous()
    B, K, T = marks.shape
    if B != lat.n_lattices:
        raise ValueError("marks must be [B, K, T]")
    tot = torch.empty((B, K), dtype=torch.float32, device=lat.device)
    end = torch.empty((B, K), dtype=torch.int32, device=lat.device)
    check(lib.nfst_score_paths(C.byref(lat.c_struct()), C.byref(sc), _ptr(marks), K, T, _ptr(tot), _ptr(end),
                               _stream()), "nfst_score_paths")
    return tot, end


def _walkers(lat: LatticeBatch, x: torch.Tensor, k: int, name: str) -> torch.Tensor:
    x = x.to(device=lat.device, dtype=torch.int64).contiguous()
    if x.shape != (lat.n_lattices * k,):
        raise ValueError(f"{name} must be [B*k] = [{lat.n_lattices * k}]")
    return x


def step(lat: LatticeBatch, state: torch.Tensor, label: torch.Tensor, k: int = 1) -> torch.Tensor:
    """state' = transition[state, label] (FSAGRUScorer.update_fsa_state, scorers.py:683-690)."""
    _need_gpu(lat)
    state, label = _walkers(lat, state, k, "state"), _walkers(lat, label, k, "label")
    out = torch.empty_like(state)
    check(lib.nfst_step(C.byref(lat.c_struct()), _ptr(state), _ptr(label), _ptr(out), int(k), _stream()), "nfst_step")
    return out


def emission_mask(lat: LatticeBatch, state: torch.Tensor, k: int = 1, inp: Optional[torch.Tensor] = None,
                  pad: int = 0, bos: int = 1, eos: int = 2, has_to_end: bool = False) -> torch.Tensor:
    """[B*k, V] mask of FSAGRUScorer.mask_out_invalid (scorers.py:1037-1054); with
    ``inp`` the bos/pad/eos legality masks (scorers.py:59-83) are fused in."""
    _need_gpu(lat)
    state = _walkers(lat, state, k, "state")
    if inp is not None:
        inp = _walkers(lat, inp, k, "inp")
    out = torch.empty((state.shape[0], lat.vocab), dtype=torch.float32, device=lat.device)
    check(lib.nfst_emission_mask(C.byref(lat.c_struct()), _ptr(state), _ptr(inp), int(pad), int(bos), int(eos),
                                 int(bool(has_to_end)), _ptr(out), int(k), _stream()), "nfst_emission_mask")
    return out


def beta_logits(lat: LatticeBatch, values: torch.Tensor, state: torch.Tensor, k: int = 1) -> torch.Tensor:
    """[B*k, V] = values[transition[state, :]] (GRUScorer beta-logit gather, scorers.py:581-593)."""
    _need_gpu(lat)
    state = _walkers(lat, state, k, "state")
    values = values.to(device=lat.device, dtype=torch.float32).contiguous().reshape(-1)
    if values.shape[0] != lat.total_rows:
        raise ValueError("values must be row-indexed [total_rows]")
    out = torch.empty((state.shape[0], lat.vocab), dtype=torch.float32, device=lat.device)
    check(lib.nfst_beta_logits(C.byref(lat.c_struct()), _ptr(values), _ptr(state), _ptr(out), int(k), _stream()),
          "nfst_beta_logits")
    return out


class ProposalStep(NamedTuple):
    symbol: torch.Tensor      # [N] int64
    logq: torch.Tensor        # [N] float32: log probability of the symbol
    logz: torch.Tensor        # [N] float32: logsumexp of the masked logits
    next_state: torch.Tensor  # [N] int64


class StepPenalties:
    """The per-walker counters and the constants of the insertion / length penalties of
    ``FSAGRUScorer.actual_left_to_right_score`` (scorers.py:654-677; defaults as its constructor, 930-933).
    ``accumulated`` [N] and ``vocab_use`` [N, V] are updated in place by every ``proposal_step``."""

    def __init__(self, n_walkers: int, vocab: int, device, insertion_mark: int, insert_threshold: int = 2,
                 insert_penalty: float = 1000.0, length_threshold: int = 0, length_penalty: float = 1000.0):
        self.accumulated = torch.zeros(n_walkers, dtype=torch.int64, device=device)
        self.vocab_use = torch.zeros(n_walkers, vocab, dtype=torch.float32, device=device)
        self.insertion_mark, self.insert_threshold, self.insert_penalty = int(insertion_mark), int(insert_threshold), float(insert_penalty)
        self.length_threshold, self.length_penalty = int(length_threshold), float(length_penalty)


class _ProposalStep(torch.autograd.Function):
    """(logq, logz) differentiable in ``scores`` (and ``values``): d logq / d scores = (onehot(symbol) -
    softmax(masked logits)) / T, as the reference's ``Categorical.log_prob`` (samplers.py:256-273)."""

    @staticmethod
    def forward(ctx, lat, scores, values, cfg):
        (state, inp, pad, bos, eos, has_to_end, temperature, uniforms, forced, extras, vstate, k, out) = cfg
        N, dev = state.shape[0], lat.device
        # (False under torch.no_grad(): a sampling loop with trainable proposal parameters then writes no [N, V] logits)
        need = ctx.needs_input_grad[1] or (values is not None and ctx.needs_input_grad[2])
        if out is not None:
            sym, logq, logz, nxt = out
        else:
            sym = torch.empty(N, dtype=torch.int64, device=dev)
            nxt = torch.empty(N, dtype=torch.int64, device=dev)
            logq = torch.empty(N, dtype=torch.float32, device=dev)
            logz = torch.empty(N, dtype=torch.float32, device=dev)
        logits = torch.empty(N, lat.vocab, dtype=torch.float32, device=dev) if need else None
        check(lib.nfst_proposal_step(C.byref(lat.c_struct()), _ptr(state), _ptr(inp), _ptr(scores), _ptr(values), int(pad),
                                     int(bos), int(eos), int(bool(has_to_end)), float(temperature), _ptr(uniforms), _ptr(forced),
                                     None if extras is None else C.byref(extras), _ptr(sym), _ptr(logq), _ptr(logz), _ptr(nxt),
                                     _ptr(logits), int(k), _stream()), "nfst_proposal_step")
        ctx.lat, ctx.k, ctx.pad, ctx.temperature = lat, k, pad, temperature
        ctx.need_values = values is not None and values.requires_grad
        ctx.n_values = 0 if values is None else values.shape[0]
        ctx.save_for_backward(logits, sym, logz, vstate)
        ctx.mark_non_differentiable(sym, nxt)
        return sym, logq, logz, nxt

    @staticmethod
    def backward(ctx, _gs, g_logq, g_logz, _gn):
        logits, sym, logz, vstate = ctx.saved_tensors
        lat = ctx.lat
        g_logq = None if g_logq is None else g_logq.to(torch.float32).contiguous()
        g_logz = None if g_logz is None else g_logz.to(torch.float32).contiguous()
        grad = torch.empty_like(logits)
        gv = torch.zeros(ctx.n_values, dtype=torch.float32, device=logits.device) if ctx.need_values else None
        check(lib.nfst_proposal_step_backward(C.byref(lat.c_struct()), _ptr(vstate), _ptr(logits), _ptr(sym), _ptr(logz),
                                              _ptr(g_logq), _ptr(g_logz), int(ctx.pad), float(ctx.temperature), _ptr(grad),
                                              _ptr(gv), int(ctx.k), _stream()), "nfst_proposal_step_backward")
        return None, grad, gv, None


def proposal_step(lat: LatticeBatch, state: torch.Tensor, scores: torch.Tensor, k: int = 1,
                  inp: Optional[torch.Tensor] = None, values: Optional[torch.Tensor] = None, pad: int = 0, bos: int = 1,
                  eos: int = 2, has_to_end: bool = False, temperature: float = 1.0,
                  uniforms: Optional[torch.Tensor] = None, forced: Optional[torch.Tensor] = None,
                  value_state: Optional[torch.Tensor] = None, penalties: Optional[StepPenalties] = None,
                  length: int = 1, out=None, not_pad: Optional[torch.Tensor] = None) -> ProposalStep:
    """One step of the reference's proposal sampler on the lattice side, fused
    (Sampler.stateful_sample, samplers.py:243-297: left_to_right_score + mask_out_invalid +
    Categorical sample / log_prob + update_fsa_state).  ``scores`` [N, V] are the proposal
    network's outputs for this step; ``values`` (row-indexed, e.g. beta) are added through the
    next-state gather of scorers.py:581-593 -- out of ``value_state`` if given (the reference reads the
    row of the state before the previous symbol was consumed), else out of ``state``; ``penalties``
    carries the insertion / length penalty counters (scorers.py:654-677) and ``length`` is the step's
    metadata["length"]; ``uniforms`` [N] drive the inverse-CDF draw, or ``forced`` [N] gives the symbols
    to evaluate.  ``logq`` and ``logz`` are differentiable in ``scores`` and ``values``.
    ``out`` = (symbol, logq, logz, next_state) tensors of a previous call or rows of buffers allocated once per
    sampling loop; ``not_pad``: an int32 device word (zeroed by the caller) that receives the number of walkers
    whose symbol is not ``pad`` (zero: every walker has ended)."""
    _need_gpu(lat)
    state = _walkers(lat, state, k, "state")
    N = state.shape[0]
    dev = lat.device
    scores = scores.to(device=dev, dtype=torch.float32).contiguous()
    if tuple(scores.shape) != (N, lat.vocab):
        raise ValueError("scores must be [n_lattices * k, vocab]")
    if inp is not None:
        inp = _walkers(lat, inp, k, "inp")
    if values is not None:
        values = values.to(device=dev, dtype=torch.float32).contiguous().reshape(-1)
        if values.shape[0] != lat.total_rows:
            raise ValueError("values must be row-indexed [total_rows]")
    if uniforms is None and forced is None:
        uniforms = torch.rand(N, device=dev)
    if uniforms is not None:
        uniforms = uniforms.to(device=dev, dtype=torch.float32).contiguous().reshape(N)
    if forced is not None:
        forced = forced.to(device=dev, dtype=torch.int64).contiguous().reshape(N)
    extras = None
    if not_pad is not None and (not_pad.dtype != torch.int32 or not_pad.numel() != 1 or not_pad.device != dev):
        raise ValueError("not_pad must be one int32 word on the batch's device")
    if out is not None:
        want = ((torch.int64, "symbol"), (torch.float32, "logq"), (torch.float32, "logz"), (torch.int64, "next_state"))
        if len(out) != 4 or any(o.dtype != d or o.shape != (N,) or o.device != dev or not o.is_contiguous() for o, (d, _) in zip(out, want)):
            raise ValueError("out must be (symbol int64, logq float32, logz float32, next_state int64), each [N] on the batch's device")
    if value_state is not None or penalties is not None or not_pad is not None:
        extras = _lib.StepExtras()
        if not_pad is not None:
            extras.not_pad = not_pad.data_ptr()
        if value_state is not None:
            value_state = _walkers(lat, value_state, k, "value_state")
            extras.value_state = value_state.data_ptr()
        if penalties is not None:
            if inp is None:
                raise ValueError("penalties need the previous symbols `inp`")
            if penalties.accumulated.shape != (N,) or penalties.vocab_use.shape != (N, lat.vocab):
                raise ValueError("penalty counters do not match the walkers")
            extras.accumulated, extras.vocab_use = penalties.accumulated.data_ptr(), penalties.vocab_use.data_ptr()
            extras.insertion_mark, extras.insert_threshold = penalties.insertion_mark, penalties.insert_threshold
            extras.insert_penalty, extras.length_threshold = penalties.insert_penalty, penalties.length_threshold
            extras.length_penalty, extras.length = penalties.length_penalty, int(length)
    vstate = value_state if value_state is not None else state
    cfg = (state, inp, pad, bos, eos, has_to_end, temperature, uniforms, forced, extras, vstate, k, out)
    return ProposalStep(*_ProposalStep.apply(lat, scores, values, cfg))


class NeuralBeta(NamedTuple):
    log_beta: torch.Tensor   # [total_rows] natural log of the reference's beta
    beta_hat: torch.Tensor   # [total_rows, H]


class _NeuralBeta(torch.autograd.Function):
    """nfst_backward_neural with nfst_backward_neural_grad behind it: differentiable in label_x
    ([V, H] = emb Wx^T + bias, made by torch ops, so Wx, bias and the embeddings get their gradients
    through it), Wh and w."""

    @staticmethod
    def forward(ctx, lat, label_x, wh, w):
        f32 = dict(device=lat.device, dtype=torch.float32)
        H = w.shape[0]
        s = lat.c_struct()
        ws = torch.empty(int(lib.nfst_neural_ws_floats(C.byref(s), H)), **f32)
        log_beta = torch.empty(lat.total_rows, **f32)
        beta_hat = torch.empty(lat.total_rows, H, **f32)
        check(lib.nfst_backward_neural(C.byref(s), label_x.data_ptr(), wh.data_ptr(), w.data_ptr(), H, log_beta.data_ptr(),
                                       beta_hat.data_ptr(), ws.data_ptr(), _stream()), "nfst_backward_neural")
        ctx.lat = lat
        ctx.save_for_backward(label_x, wh, w, beta_hat, ws)
        return log_beta, beta_hat

    @staticmethod
    def backward(ctx, g_log_beta, g_beta_hat):
        lat = ctx.lat
        label_x, wh, w, beta_hat, ws_fwd = ctx.saved_tensors
        f32 = dict(device=lat.device, dtype=torch.float32)
        H = w.shape[0]
        s = lat.c_struct()
        g_lb = (torch.zeros(lat.total_rows, **f32) if g_log_beta is None else g_log_beta.to(**f32)).contiguous()
        # rows the sweep never reached carry -inf and no gradient
        g_lb = torch.where(torch.isfinite(g_lb), g_lb, torch.zeros_like(g_lb))
        g_bh = None if g_beta_hat is None else g_beta_hat.to(**f32).contiguous()
        gamma = torch.zeros(lat.total_rows, H, **f32)
        g_x = torch.zeros_like(label_x)
        g_w = torch.zeros(H, **f32)
        ws = torch.empty(int(lib.nfst_neural_grad_ws_floats(C.byref(s), H)), **f32)
        wh_t = wh.t().contiguous()
        check(lib.nfst_backward_neural_grad(C.byref(s), label_x.data_ptr(), wh_t.data_ptr(), w.data_ptr(), H,
                                            beta_hat.data_ptr(), ws_fwd.data_ptr(), g_lb.data_ptr(),
                                            0 if g_bh is None else g_bh.data_ptr(), gamma.data_ptr(), g_x.data_ptr(),
                                            g_w.data_ptr(), ws.data_ptr(), _stream()), "nfst_backward_neural_grad")
        # dL/dWh[i, j] = sum over states of gamma(s)[i] beta_hat(s)[j]: a library GEMM
        return None, g_x, _gram(gamma, beta_hat), g_w


def _gram(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a^T b for tall, narrow a and b ([n, H], n ~ 5e5): as one GEMM with K = n the library takes 0.3 ms at
    H = 8 ... 1.25 ms at H = 256 on the BASELINE batch's rows; 256 row chunks as one batched GEMM and a sum take
    0.04 ... 0.54 ms (`profiles/tune/gemm_tall_skinny.py`) and add in a better order."""
    n, H = a.shape
    chunks = 256 if n >= 256 * 64 else 1
    m = (n // chunks) * chunks
    out = torch.bmm(a[:m].view(chunks, -1, H).transpose(1, 2), b[:m].view(chunks, -1, H)).sum(0)
    if m < n:
        out = out + a[m:].t() @ b[m:]
    return out


def backward_neural(lat: LatticeBatch, emb: torch.Tensor, Wx: torch.Tensor, Wh: torch.Tensor, W: torch.Tensor,
                    bias: torch.Tensor) -> NeuralBeta:
    """``FSAGRUScorer.compute_beta_per_sample`` / ``compute_beta_parallel`` with their Tree-LSTM-style
    messages (scorers.py:692-751, 753-856), parameters named as there: ``emb`` [V, H] mark embeddings,
    ``Wx``, ``Wh`` [H, H], ``W`` [1, H] or [H], ``bias`` [H].  The per-label part ``Wx e(l) + bias`` is
    one [V, H] x [H, H] product made here (a plain library GEMM); everything that depends on the
    lattice runs in the kernel.  Differentiable in all five parameters (the reference's
    ``tune_proposal`` trains them through ``compute_beta``, lightning.py:339-406): the gradient is
    ``nfst_backward_neural_grad`` plus two library GEMMs."""
    _need_gpu(lat)
    f32 = dict(device=lat.device, dtype=torch.float32)
    emb, Wx, Wh, bias = emb.to(**f32), Wx.to(**f32), Wh.to(**f32), bias.to(**f32)
    w = W.to(**f32).reshape(-1).contiguous()
    H = w.shape[0]
    if emb.shape != (lat.vocab, H) or Wx.shape != (H, H) or Wh.shape != (H, H) or bias.shape != (H,):
        raise ValueError("emb must be [V, H], Wx and Wh [H, H], W [H] and bias [H]")
    label_x = torch.addmm(bias, emb, Wx.t()).contiguous()
    return NeuralBeta(*_NeuralBeta.apply(lat, label_x, Wh.contiguous(), w))


def gather_label_scores(lat: LatticeBatch, theta, arc_scores=None) -> torch.Tensor:
    """Per-arc log weights in canonical order (WFSTScorer, scorers.py:1671-1687)."""
    _need_gpu(lat)
    sc, keep = _scores(lat, theta, arc_scores)
    out = torch.empty(lat.total_arcs, dtype=torch.float32, device=lat.device)
    check(lib.nfst_gather_label_scores(C.byref(lat.c_struct()), C.byref(sc), _ptr(out), _stream()),
          "nfst_gather_label_scores")
    return out


MASK_STATICRNN, MASK_GPT2 = 0, 1


def _plp_args(pad, bos, eos, max_length, temp, normalize, smoothing, mask_mode):
    return (int(pad), int(bos), int(eos), -1 if max_length is None else int(max_length), C.c_float(temp),
            int(bool(normalize)), C.c_float(smoothing), int(mask_mode))


class _PathLogprob(torch.autograd.Function):
    """out[n] = sum_t (log_softmax row gathered / label-smoothed), d out / d scores by
    nfst_path_logprob_backward (recomputes the rows; nothing but the inputs is saved)."""

    @staticmethod
    def forward(ctx, scores, marks, cfg):
        N, T, V = scores.shape
        out = torch.empty(N, dtype=torch.float32, device=scores.device)
        check(lib.nfst_path_logprob(_ptr(scores), _ptr(marks), N, T, V, *_plp_args(*cfg), _ptr(out), _stream()),
              "nfst_path_logprob")
        ctx.cfg = cfg
        ctx.save_for_backward(scores, marks)
        return out

    @staticmethod
    def backward(ctx, g):
        scores, marks = ctx.saved_tensors
        N, T, V = scores.shape
        g = g.to(torch.float32).contiguous()
        grad = torch.empty_like(scores)
        check(lib.nfst_path_logprob_backward(_ptr(scores), _ptr(marks), _ptr(g), N, T, V, *_plp_args(*ctx.cfg), _ptr(grad),
                                             _stream()), "nfst_path_logprob_backward")
        return grad, None, None


def path_logprob(scores: torch.Tensor, marks: torch.Tensor, pad: int = 0, bos: int = 1, eos: int = 2,
                 max_length: Optional[int] = None, temp: float = 1.0, normalize: bool = True,
                 smoothing: float = 0.0, mask_mode: int = MASK_STATICRNN) -> torch.Tensor:
    """Fused masks + log_softmax + gather + pad-masked sum over time
    (StaticRNNScorer.evaluate_seq_with_temp, scorers.py:1564-1611): scores [N,T,V], marks [N,T] -> [N].
    ``smoothing > 0`` selects the training branch (label-smoothed target, scorers.py:1584-1592).
    Differentiable with respect to ``scores`` -- the reference trains p~ through this op
    (lightning.py:511-516)."""
    if scores.device.type != "cuda":
        raise RuntimeError("nfst_amd: path_logprob runs on the MI355X only (no CPU fallback)")
    scores = scores.to(torch.float32).contiguous()
    marks = marks.to(device=scores.device, dtype=torch.int64).contiguous()
    N, T, V = scores.shape
    if marks.shape != (N, T):
        raise ValueError("marks must be [N, T]")
    return _PathLogprob.apply(scores, marks, (pad, bos, eos, max_length, temp, normalize, smoothing, mask_mode))


def gpt2_logprob(logits: torch.Tensor, x: torch.Tensor, pad: int = 0) -> torch.Tensor:
    """The arithmetic of ``GPT2Wrapper.forward`` after the language model (modules/transformer.py:45-52):
    ``logits`` [N, T+1, V] are the model's outputs for the bos-shifted input ``cat(bos, x)``, ``x`` [N, T] the
    marks; gold = ``cat(x, pad)``; the pad logit becomes -1e8, log_softmax, gather of gold, positions
    holding pad contribute 0, sum over time -> [N].  Differentiable with respect to ``logits``."""
    N, T = x.shape
    if tuple(logits.shape[:2]) != (N, T + 1):
        raise ValueError("logits must be [N, T + 1, V] for x [N, T]")
    gold = torch.cat((x, x.new_full((N, 1), pad)), dim=1)
    return path_logprob(logits, gold, pad=pad, normalize=True, mask_mode=MASK_GPT2)


def iwae(log_p: torch.Tensor, log_q: torch.Tensor):
    """(log_marginal [B], log_w [B,K]) of Estimators.iwae (modules/estimatros.py:11-44)."""
    if log_p.device.type != "cuda":
        raise RuntimeError("nfst_amd: iwae runs on the MI355X only (no CPU fallback)")
    log_p = log_p.to(torch.float32).contiguous()
    log_q = log_q.to(device=log_p.device, dtype=torch.float32).contiguous()
    B, K = log_p.shape
    log_w = torch.empty_like(log_p)
    lm = torch.empty(B, dtype=torch.float32, device=log_p.device)
    check(lib.nfst_iwae(_ptr(log_p), _ptr(log_q), B, K, _ptr(log_w), _ptr(lm), _stream()), "nfst_iwae")
    return lm, log_w
