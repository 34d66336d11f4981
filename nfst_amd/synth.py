"""Deterministic synthetic lattices in the reference's dense table format.

The reference ships no data, so every lattice used by the
tests and by ``bench.py`` comes from here.  A lattice is a deterministic acyclic
mark-FSA in the encoding that ``FSAGRUScorer.get_state_mask_pynini`` produces
(/root/reference/src/modules/scorers.py:995-1035):

* ``emission  [S+1, V]`` bool  -- arc with label ``l`` leaves state ``s``
* ``transition[S+1, V]`` int64 -- its destination (0 where there is no arc)
* state 0 is the start, row ``S`` is the sink with a ``pad`` self loop.

Two shapes are generated:

``layered_lattice``  the BASELINE workload: ~2k states,
    ~20k arcs, state ids randomly permuted, at least one pair of parallel arcs.
``edit_lattice``     a transliteration-shaped lattice: the x o T o y edit grid of
    /root/reference/src/fsm/tr.py:321-390 projected on its mark tape
    (insertion = [output-mark y], deletion = [input-mark x],
    substitution = [insertion-mark input-mark x output-mark y]).

Only numpy is used so that the generator is importable everywhere (tests, the
oracle checks, bench.py) and is bit-reproducible from the seed.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

PAD, BOS, EOS = 0, 1, 2  # the reference scorers' defaults (scorers.py:475-477)
N_SPECIAL = 3


@dataclass
class SynthLattice:
    """One lattice as an arc list sorted by (src, label) -- the canonical order."""

    n_rows: int  # S + 1 (row S is the sink)
    vocab: int
    src: np.ndarray  # int32 [A]   (includes the sink's pad self loop, last)
    label: np.ndarray  # int32 [A]
    dst: np.ndarray  # int32 [A]
    weight: Optional[np.ndarray] = None  # float32 [A] (weighted emission) or None

    @property
    def n_arcs(self) -> int:
        return int(self.src.shape[0])

    def dense(self, weighted: bool = False) -> Tuple[np.ndarray, np.ndarray]:
        """(emission, transition) exactly as scorers.py:1006-1032 lays them out."""
        if weighted:
            emission = np.full((self.n_rows, self.vocab), -np.inf, dtype=np.float32)
            w = self.weight if self.weight is not None else np.zeros(self.n_arcs, np.float32)
            emission[self.src, self.label] = w
        else:
            emission = np.zeros((self.n_rows, self.vocab), dtype=np.bool_)
            emission[self.src, self.label] = True
        transition = np.zeros((self.n_rows, self.vocab), dtype=np.int64)
        transition[self.src, self.label] = self.dst
        return emission, transition


def _finish(n_rows: int, vocab: int, src, label, dst, weight=None) -> SynthLattice:
    """Append the sink's pad self loop and sort arcs by (src, label)."""
    sink = n_rows - 1
    src = np.concatenate([np.asarray(src, np.int64), [sink]])
    label = np.concatenate([np.asarray(label, np.int64), [PAD]])
    dst = np.concatenate([np.asarray(dst, np.int64), [sink]])
    if weight is not None:
        weight = np.concatenate([np.asarray(weight, np.float32), np.zeros(1, np.float32)])
    order = np.lexsort((label, src))
    key = src[order] * vocab + label[order]
    assert np.all(np.diff(key) > 0), "lattice is not deterministic"
    return SynthLattice(
        n_rows=n_rows,
        vocab=vocab,
        src=src[order].astype(np.int32),
        label=label[order].astype(np.int32),
        dst=dst[order].astype(np.int32),
        weight=None if weight is None else weight[order].astype(np.float32),
    )


def _units_mod(m: int) -> np.ndarray:
    return np.array([u for u in range(1, m) if np.gcd(u, m) == 1], dtype=np.int64)


def layered_lattice(
    seed: int,
    n_states: int = 2000,
    avg_degree: float = 10.0,
    vocab: int = 256,
    width: int = 16,
    span: int = 8,
    max_degree: int = 24,
    permute: bool = True,
    weighted: bool = False,
) -> SynthLattice:
    """A layered DAG lattice.

    State 0 --bos--> state 1; the remaining real states are laid out in layers of
    ``width`` states; a state in layer ``l`` sends ~Poisson(avg_degree) arcs with
    distinct labels to uniformly chosen states of layers ``l+1 .. l+span``; every
    state has an arc from the previous layer and one to the next layer, so every
    state is reachable and co-reachable and the longest path visits every layer;
    the last layer leaves by ``eos`` to the sink (row ``n_states``), which carries
    the ``pad`` self loop.  ``width=1`` gives the near-sequential worst case
    (destinations within the next ``span`` ids).  Ids 1..S-1 are then permuted.
    """
    assert n_states >= 4 and vocab > N_SPECIAL + 8 and width >= 1 and span >= 1
    rng = np.random.default_rng(seed)
    S = int(n_states)
    sink = S
    nlab = vocab - N_SPECIAL
    # layer of each real state: 0 -> layer 0, 1 -> layer 1, the rest in chunks
    layer = np.zeros(S, dtype=np.int64)
    layer[1] = 1
    if S > 2:
        layer[2:] = 2 + (np.arange(S - 2) // width)
    L = int(layer[-1]) + 1  # number of layers
    first = np.searchsorted(layer, np.arange(L), side="left")  # first state of layer
    count = np.bincount(layer, minlength=L)

    inner = np.nonzero(layer < L - 1)[0]  # states with ordinary out arcs
    inner = inner[inner >= 1]
    # coverage arcs: every state t of layer l>=2 gets one arc from layer l-1
    tgt = np.arange(2, S, dtype=np.int64)
    lt = layer[tgt]
    cov_src = first[lt - 1] + rng.integers(0, 1 << 30, size=tgt.shape[0]) % count[lt - 1]
    cov_cnt = np.bincount(cov_src, minlength=S)

    deg = np.clip(rng.poisson(avg_degree, size=S), 1, max_degree).astype(np.int64)
    deg = np.minimum(deg, nlab)
    rnd_cnt = np.maximum(deg - cov_cnt, 0)
    rnd_cnt[cov_cnt == 0] = np.maximum(rnd_cnt[cov_cnt == 0], 1)
    rnd_cnt[layer >= L - 1] = 0
    rnd_cnt[0] = 0
    r_src = np.repeat(np.arange(S, dtype=np.int64), rnd_cnt)
    r_first = np.cumsum(rnd_cnt) - rnd_cnt  # offset of each state's first random arc
    r_idx = np.arange(r_src.shape[0]) - r_first[r_src]
    hop = 1 + rng.integers(0, span, size=r_src.shape[0])
    # a state without coverage arcs sends its first arc to the next layer
    force_next = (r_idx == 0) & (cov_cnt[r_src] == 0)
    hop[force_next] = 1
    dl = np.minimum(layer[r_src] + hop, L - 1)
    r_dst = first[dl] + rng.integers(0, 1 << 30, size=r_src.shape[0]) % count[dl]
    # one guaranteed pair of parallel arcs: second random arc copies the first
    cand = np.nonzero(rnd_cnt >= 2)[0]
    if cand.shape[0] > 0:
        s0 = int(cand[rng.integers(0, cand.shape[0])])
        r_dst[r_first[s0] + 1] = r_dst[r_first[s0]]

    src = np.concatenate([cov_src, r_src])
    dst = np.concatenate([tgt, r_dst])
    # distinct labels per state: arithmetic progression modulo nlab
    order = np.argsort(src, kind="stable")
    src, dst = src[order], dst[order]
    tot = np.bincount(src, minlength=S)
    assert tot.max() <= nlab
    j = np.arange(src.shape[0]) - (np.cumsum(tot) - tot)[src]
    units = _units_mod(nlab)
    base = rng.integers(0, nlab, size=S)
    stride = units[rng.integers(0, units.shape[0], size=S)]
    label = N_SPECIAL + (base[src] + stride[src] * j) % nlab

    # bos arc 0 -> 1 and eos arcs last layer -> sink
    last = np.nonzero(layer == L - 1)[0]
    src = np.concatenate([[0], src, last])
    dst = np.concatenate([[1], dst, np.full(last.shape[0], sink)])
    label = np.concatenate([[BOS], label, np.full(last.shape[0], EOS)])

    if permute and S > 2:
        perm = np.arange(S + 1, dtype=np.int64)
        perm[1:S] = 1 + rng.permutation(S - 1)
        src, dst = perm[src], perm[dst]
    weight = None
    if weighted:
        weight = rng.normal(-0.5, 0.25, size=src.shape[0]).astype(np.float32)
    return _finish(S + 1, vocab, src, label, dst, weight)


def edit_lattice(
    x: Sequence[int],
    y: Sequence[int],
    vocab: int,
    input_mark: int = 3,
    output_mark: int = 4,
    insertion_mark: int = 5,
    substitutions: bool = True,
    seed: Optional[int] = None,
) -> SynthLattice:
    """Transliteration-shaped lattice: the edit grid of x and y on the mark tape.

    Node (i, j) has consumed x[:i] and produced y[:j].  Edits and the marks they
    emit follow /root/reference/src/fsm/tr.py:329-388:
      insertion     (i,j)->(i,j+1)    [output-mark, y_j]
      deletion      (i,j)->(i+1,j)    [input-mark, x_i]
      substitution  (i,j)->(i+1,j+1)  [insertion-mark, input-mark, x_i, output-mark, y_j]
    wrapped as bos . M . eos . pad* (path_semiring.py:245-265).  Each mark of an
    edit is one arc through a private chain state, so the machine is
    deterministic.  ``seed`` permutes the state ids.
    """
    n, m = len(x), len(y)
    src: List[int] = []
    lab: List[int] = []
    dst: List[int] = []
    nxt = [1]

    def new_state() -> int:
        nxt[0] += 1
        return nxt[0] - 1

    node = {}
    for i in range(n + 1):
        for j in range(m + 1):
            node[(i, j)] = new_state()

    def chain(a: int, marks: Sequence[int], b: int) -> None:
        cur = a
        for k, mk in enumerate(marks):
            nx = b if k == len(marks) - 1 else new_state()
            src.append(cur), lab.append(int(mk)), dst.append(nx)
            cur = nx

    src.append(0), lab.append(BOS), dst.append(node[(0, 0)])
    for i in range(n + 1):
        for j in range(m + 1):
            a = node[(i, j)]
            if j < m:
                chain(a, [output_mark, y[j]], node[(i, j + 1)])
            if i < n:
                chain(a, [input_mark, x[i]], node[(i + 1, j)])
            if substitutions and i < n and j < m:
                chain(a, [insertion_mark, input_mark, x[i], output_mark, y[j]], node[(i + 1, j + 1)])
    S = nxt[0]
    sink = S
    src.append(node[(n, m)]), lab.append(EOS), dst.append(sink)
    src_a, dst_a = np.asarray(src, np.int64), np.asarray(dst, np.int64)
    if seed is not None and S > 2:
        rng = np.random.default_rng(seed)
        perm = np.arange(S + 1, dtype=np.int64)
        perm[1:S] = 1 + rng.permutation(S - 1)
        src_a, dst_a = perm[src_a], perm[dst_a]
    return _finish(S + 1, vocab, src_a, np.asarray(lab, np.int64), dst_a)


def label_scores(seed: int, vocab: int, mean: float = -2.3, std: float = 0.5) -> np.ndarray:
    """theta ~ N(mean, std^2) per label, float32."""
    rng = np.random.default_rng(seed)
    return rng.normal(mean, std, size=vocab).astype(np.float32)


def collate_dense(tables: Sequence[Tuple[np.ndarray, np.ndarray]], pad: int = PAD):
    """Pad per-lattice tables to the batch maximum the way the reference's
    ``T9FSADataModule.collate`` does (dataset_reader.py:175-186 via
    ``Utils.pad_sequence``, preprocess_util.py:368-392): every array is padded
    along dim 0 with the *pad id* -- bool rows become ``bool(pad)`` and
    transition rows become ``pad``."""
    n = max(t[0].shape[0] for t in tables)
    em0, tr0 = tables[0]
    em = np.full((len(tables), n) + em0.shape[1:], fill_value=pad, dtype=em0.dtype)
    tr = np.full((len(tables), n) + tr0.shape[1:], fill_value=pad, dtype=tr0.dtype)
    for i, (e, t) in enumerate(tables):
        em[i, : e.shape[0]] = e
        tr[i, : t.shape[0]] = t
    return em, tr


def batch_arcs(lattices: Sequence[SynthLattice]):
    """Concatenate arc lists: (n_rows[B], arc_off[B+1], src, label, dst, weight|None)."""
    n_rows = np.array([l.n_rows for l in lattices], dtype=np.int32)
    arc_off = np.zeros(len(lattices) + 1, dtype=np.int64)
    arc_off[1:] = np.cumsum([l.n_arcs for l in lattices])
    src = np.concatenate([l.src for l in lattices]).astype(np.int32)
    label = np.concatenate([l.label for l in lattices]).astype(np.int32)
    dst = np.concatenate([l.dst for l in lattices]).astype(np.int32)
    weight = None
    if all(l.weight is not None for l in lattices):
        weight = np.concatenate([l.weight for l in lattices]).astype(np.float32)
    return n_rows, arc_off, src, label, dst, weight


def snips_shaped_batch(n_lattices: int = 64, vocab: int = 250, first_seed: int = 3000) -> List[SynthLattice]:
    """BASELINE configs[2]-shaped lattices: tagging machines are long and narrow -- a few tag states
    per token position, up to ~750 positions (S ~ 400..1500, V ~ 250).  (The real SNIPS machines are
    built with mfst / OpenFST, which this image does not have: SURVEY.md section 8c.)"""
    rng = np.random.default_rng(64)
    return [layered_lattice(first_seed + i, n_states=int(rng.integers(400, 1501)), avg_degree=float(rng.choice([3.0, 5.0, 8.0])),
                            vocab=vocab, width=int(rng.choice([2, 3, 6, 8])), span=int(rng.choice([1, 2])), max_degree=40)
            for i in range(n_lattices)]


def without_parallel_arcs(l: SynthLattice) -> SynthLattice:
    """The lattice with one arc kept per (src, dst) pair (the one with the smallest label): the
    reference's ``compute_beta_parallel`` never releases a state with two arcs to the same
    destination, so its algorithm can only be timed on such lattices."""
    nl = l.src != l.dst
    key = l.src[nl].astype(np.int64) * l.n_rows + l.dst[nl]
    _, first = np.unique(key, return_index=True)
    first.sort()
    w = None if l.weight is None else l.weight[nl][first]
    return _finish(l.n_rows, l.vocab, l.src[nl][first], l.label[nl][first], l.dst[nl][first], w)


def bench_batch(
    n_lattices: int, first_seed: int = 1234, n_states: Optional[int] = None, **kw
) -> List[SynthLattice]:
    """The BASELINE batch: lattice i uses seed ``first_seed + i``; unless
    ``n_states`` is given, S ~ U[1800, 2200] drawn from the lattice's own seed."""
    out = []
    for i in range(n_lattices):
        seed = first_seed + i
        S = n_states
        if S is None:
            S = int(np.random.default_rng(seed ^ 0x5EED).integers(1800, 2201))
        out.append(layered_lattice(seed, n_states=S, **kw))
    return out
