"""The chunked programs for deep, narrow lattices (nfst_pack_chunks, include/nfst_hip.h) checked on the host: the three passes
of chunk_kernels.h replayed in numpy float64 from the packed arrays, against the oracle's forward-backward.  (The kernels
themselves: tests/test_gpu_chunks.py.)"""
import numpy as np
import pytest

from nfst_amd import _lib, synth
from nfst_amd.lattice import ChunkProgram, LatticeBatch
from oracle import oracle as O

LAST, ZERO = 0x40, 0x80


def replay(lat: LatticeBatch, ck: ChunkProgram, b: int, d: int, weight: np.ndarray) -> np.ndarray:
    """values (linear, float64) of every row of lattice b in direction d (0 alpha, 1 beta), the way the kernel computes them;
    `weight` per canonical arc of the lattice.  Small lattices only: plain floats, no exponents."""
    C_, F, R, npos, tab_off, s_off, p_off, _ = (int(x) for x in ck.meta_host[b, d])
    tab = ck._t["tab"].numpy().reshape(-1, 4)[tab_off:tab_off + C_]
    stream = ck._t["stream"].numpy().view(np.uint32)
    pos = ck._t["pos"].numpy()[p_off:p_off + npos]
    T = np.zeros((npos, F))
    rings = []
    for c in range(C_):  # pass 1
        a_c, begin, cnt = int(tab[c, 0]), int(tab[c, 1]), int(tab[c, 2])
        ring = np.zeros((R, F))
        for f in range(F):
            q = a_c - 1 - f
            if q >= 0:
                ring[q & (R - 1), f] = 1.0
        acc, p = np.zeros(F), a_c
        for k in range(cnt):
            e = int(stream[s_off + begin + k])
            w = 0.0 if e & ZERO else weight[e >> 8]
            acc = acc + ring[e & 63] * w
            if e & LAST:
                ring[p & (R - 1)] = acc
                T[p] = acc
                acc, p = np.zeros(F), p + 1
        assert p == (int(tab[c + 1, 0]) if c + 1 < C_ else npos)
        rings.append(ring)
    front = np.zeros((C_, F))  # pass 2
    front[0, 0] = 1.0
    for c in range(C_ - 1):
        a_c, a_n = int(tab[c, 0]), int(tab[c + 1, 0])
        for i in range(F):
            p = a_n - 1 - i
            if p >= a_c:
                front[c + 1, i] = rings[c][p & (R - 1)] @ front[c]
            elif p >= 0:
                front[c + 1, i] = front[c, a_c - 1 - p]
    val = np.zeros(int(lat.n_rows[b]))  # pass 3
    val[pos[0]] = 1.0
    for c in range(C_):
        a_c, a_n = int(tab[c, 0]), (int(tab[c + 1, 0]) if c + 1 < C_ else npos)
        for p in range(a_c, a_n):
            val[pos[p]] = T[p] @ front[c]
    return val


def check_structure(lat: LatticeBatch, ck: ChunkProgram):
    h = ck._h
    assert h["n_lattices"] == lat.n_lattices and h["total_rows"] == lat.total_rows and h["total_arcs"] == lat.total_arcs
    stream = ck._t["stream"].numpy().view(np.uint32)
    src, dst = lat.arc_src.numpy(), lat.arc_dst.numpy()
    for b in range(lat.n_lattices):
        a0, na = int(lat.arc_off[b]), int(lat.n_arcs[b])
        dp = np.flatnonzero(src[a0:a0 + na] != dst[a0:a0 + na])
        for d in (0, 1):
            C_, F, R, npos, tab_off, s_off, p_off, t_off = (int(x) for x in ck.meta_host[b, d])
            assert 1 <= C_ and C_ * F <= h["threads"] and F <= R <= 64 and (R & (R - 1)) == 0
            assert npos == int(lat.meta_host[b, _lib.META_N_REACH])
            tab = ck._t["tab"].numpy().reshape(-1, 4)[tab_off:tab_off + C_]
            assert tab[0, 0] == 1 and np.all(np.diff(tab[:, 0]) > 0) and tab[0, 1] == 0
            assert np.all(tab[1:, 1] == np.cumsum(tab[:-1, 2]))
            n_e = int(tab[-1, 1] + tab[-1, 2])
            e = stream[s_off:s_off + n_e]
            arcs = (e >> 8)[(e & ZERO) == 0]
            assert np.array_equal(np.sort(arcs), dp), "every arc of the lattice exactly once"
            lab = ck._t["label"].numpy().view(np.uint16)[s_off:s_off + n_e]
            assert np.array_equal(lab[(e & ZERO) == 0], lat.arc_label.numpy()[a0 + arcs]), "the label beside every entry"
            assert int(((e & LAST) != 0).sum()) == npos - 1
            pos = ck._t["pos"].numpy()[p_off:p_off + npos]
            assert len(set(pos.tolist())) == npos and pos[0] == (0 if d == 0 else int(lat.sink[b]))
            # operands lie below their state, at most R - 1 positions back, and a chunk reaches at most F positions below its start
            posof = np.full(int(lat.n_rows[b]), -1); posof[pos] = np.arange(npos)
            # (entry k adds to the state after as many states as there are LAST flags before k; the zero-weight entries that pad
            # a chunk to a multiple of eight follow its last state)
            state_of_entry = pos[1:][np.minimum(np.cumsum(np.concatenate([[0], (e & LAST) != 0]))[:-1], npos - 2)]
            assert np.all(tab[:, 2] % 8 == 0)
            for c in range(C_):
                lo, hi = int(tab[c, 1]), int(tab[c, 1] + tab[c, 2])
                ee, ss = e[lo:hi], state_of_entry[lo:hi]
                live = (ee & ZERO) == 0
                a = a0 + (ee >> 8)[live]
                st = ss[live]
                assert np.array_equal(st, (dst if d == 0 else src)[a]), "entries are grouped by the state they add to"
                q = posof[(src if d == 0 else dst)[a]]
                p = posof[st]
                assert np.all(q < p) and np.all(p - q < R) and np.array_equal(q & (R - 1), ee[live] & 63)
                assert np.all(q >= int(tab[c, 0]) - F)


def oracle_values(l: synth.SynthLattice, theta: np.ndarray):
    sc = theta[l.label].astype(np.float64) + (0.0 if l.weight is None else l.weight.astype(np.float64))
    return sc, O.forward_backward(l.n_rows, l.src, l.dst, sc)


@pytest.mark.parametrize("opts", [dict(), dict(threads=64), dict(threads=256, max_chunks=3), dict(max_chunks=1), dict(lds_bytes=8192, threads=128)])
def test_chunked_programs_replayed_on_the_host_match_the_oracle(opts):
    lats = [synth.layered_lattice(100 + i, n_states=n, avg_degree=deg, vocab=40, width=w, span=sp, max_degree=12, weighted=bool(i & 1))
            for i, (n, deg, w, sp) in enumerate([(60, 3.0, 2, 1), (90, 4.0, 3, 2), (40, 2.0, 1, 1), (120, 5.0, 4, 2), (30, 3.0, 2, 3),
                                                 (7, 2.0, 2, 1), (200, 3.0, 6, 1)])]
    for l in lats:
        if l.weight is None:
            l.weight = np.zeros(l.src.shape[0], np.float32)
    lat = LatticeBatch.from_synth(lats)
    assert lat.build_chunks(force=True, **opts)
    ck = lat.chunks
    check_structure(lat, ck)
    theta = synth.label_scores(7, 40, mean=-0.3, std=0.5)
    for b, l in enumerate(lats):
        sc, o = oracle_values(l, theta)
        a0, na = int(lat.arc_off[b]), int(lat.n_arcs[b])
        # canonical arcs are the lattice's arcs sorted by (src, label): weights in that order
        order = np.lexsort((l.label, l.src))
        assert np.array_equal(lat.arc_src.numpy()[a0:a0 + na], l.src[order])
        w = np.exp(sc[order])
        for d, key in ((0, "logalpha"), (1, "logbeta")):
            val = replay(lat, ck, b, d, w)
            with np.errstate(divide="ignore"):
                got = np.log(val)
            ref = o[key]
            fin = np.isfinite(ref)
            assert np.array_equal(np.isfinite(got), fin)
            assert np.max(np.abs(got[fin] - ref[fin])) <= 1e-9, (b, d)
        assert abs(np.log(replay(lat, ck, b, 1, w)[0]) - o["logZ"]) <= 1e-9


def test_which_batches_get_chunked_programs():
    """deep and narrow: yes (by the cost model); the BASELINE shape: no, not even forced (its arcs reach too far back)"""
    snips = LatticeBatch.from_synth(synth.snips_shaped_batch(16))
    assert snips.build_chunks() and snips.chunks is not None
    check_structure(snips, snips.chunks)
    m = snips.chunks.meta_host
    assert m[:, :, _lib.CHK_C].max() <= 128  # (pass 2 is a chain of C steps: the packer balances it against pass 1)
    base = LatticeBatch.from_synth(synth.bench_batch(4))
    assert not base.build_chunks() and base.chunks is None
    assert not base.build_chunks(force=True)
    wide = LatticeBatch.from_synth([synth.layered_lattice(5, n_states=300, avg_degree=6.0, vocab=64, width=48, span=1, max_degree=24)])
    assert not wide.build_chunks(force=True)  # levels of 48 states: an arc reaches more than 63 positions back
    with pytest.raises(_lib.NfstError):
        ChunkProgram.build(snips, threads=100)  # not a multiple of 64


@pytest.mark.parametrize("seed", list(range(12)))
def test_random_cuts_replayed_on_the_host(seed):
    """random narrow lattices, random workgroup sizes / LDS budgets / chunk limits: the structure of every program and the replayed
    passes against the oracle (the host-side twin of test_gpu_chunks.test_fuzz_chunked_flavour_against_oracle)"""
    rng = np.random.default_rng(4000 + seed)
    V = int(rng.choice([24, 64]))
    lats = []
    while len(lats) < int(rng.integers(1, 6)):
        try:
            lats.append(synth.layered_lattice(int(rng.integers(1, 1 << 30)), n_states=int(rng.choice([4, 9, 30, 80, 200])),
                                              avg_degree=float(rng.choice([1.5, 3.0, 5.0])), vocab=V, width=int(rng.choice([1, 2, 3, 5, 8])),
                                              span=int(rng.choice([1, 2, 3])), max_degree=min(10, (V - 12) // 2)))
        except AssertionError:
            continue
    lat = LatticeBatch.from_synth(lats)
    opts = dict(threads=int(rng.choice([64, 128, 512, 1024])), max_chunks=int(rng.choice([0, 0, 1, 2, 5])), lds_bytes=int(rng.choice([0, 8192, 32768])))
    if not lat.build_chunks(force=True, **opts):
        pytest.skip("no cut within these limits")
    check_structure(lat, lat.chunks)
    theta = synth.label_scores(seed, V, mean=-0.2, std=0.6)
    for b, l in enumerate(lats):
        sc, o = oracle_values(l, theta)
        w = np.exp(sc[np.lexsort((l.label, l.src))])
        for d, key in ((0, "logalpha"), (1, "logbeta")):
            with np.errstate(divide="ignore"):
                got = np.log(replay(lat, lat.chunks, b, d, w))
            fin = np.isfinite(o[key])
            assert np.array_equal(np.isfinite(got), fin) and np.max(np.abs(got[fin] - o[key][fin])) <= 1e-9, (b, d)
