"""Host packer (nfst_amd/csrc/pack.cpp) and C-ABI surface -- CPU only."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from oracle import oracle as O
from nfst_amd import _lib, synth
from nfst_amd.lattice import LatticeBatch
from tests.stream_check import replay

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "nfst_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(nfst_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 18
    raw = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in nfst_hip.h but not exported"
    assert declared == set(_lib.EXPORTS)
    assert _lib.lib.nfst_abi_version() == _lib.header_abi_version() == _lib.ABI_VERSION


def test_ops_fail_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from nfst_amd import ops
    lat = LatticeBatch.from_synth([synth.layered_lattice(1, n_states=20, avg_degree=3, vocab=32, width=3, span=2)])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.forward_backward(lat, torch.zeros(32))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.path_logprob(torch.zeros(1, 2, 8), torch.zeros(1, 2, dtype=torch.long))


def _cases():
    return [
        synth.layered_lattice(3, n_states=30, avg_degree=3.0, vocab=40, width=4, span=2),
        synth.layered_lattice(4, n_states=200, avg_degree=8.0, vocab=64, width=9, span=5),
        synth.layered_lattice(5, n_states=64, avg_degree=5.0, vocab=48, width=1, span=6),  # near-sequential
        synth.edit_lattice([10, 11, 12, 13], [20, 21, 22], vocab=40, seed=2),
        synth.layered_lattice(6, n_states=150, avg_degree=6.0, vocab=64, width=7, span=3, weighted=True),
    ]


@pytest.mark.parametrize("pad", [0, 7])
def test_dense_pack_matches_oracle_arcs(pad):
    lats = _cases()[:4]
    em, tr = synth.collate_dense([l.dense() for l in [synth.layered_lattice(3, n_states=30, avg_degree=3.0, vocab=40, width=4, span=2),
                                                      synth.edit_lattice([10, 11, 12, 13], [20, 21, 22], vocab=40, seed=2)]], pad=pad)
    lat = LatticeBatch.from_dense(em, tr)
    assert lat.uniform_rows and lat.n_rows[0] == em.shape[1]
    for b in range(em.shape[0]):
        src, label, dst, _ = O.dense_to_arcs(em[b], tr[b])
        a0, n = int(lat.arc_off[b]), int(lat.n_arcs[b])
        assert n == src.shape[0]
        assert np.array_equal(lat.arc_src.numpy()[a0:a0 + n], src)
        assert np.array_equal(lat.arc_label.numpy()[a0:a0 + n], label)
        assert np.array_equal(lat.arc_dst.numpy()[a0:a0 + n], dst)
        rp = lat.row_ptr.numpy()[int(lat.row_off[b]) + b: int(lat.row_off[b]) + b + int(lat.n_rows[b]) + 1]
        assert rp[0] == a0 and rp[-1] == a0 + n and np.all(np.diff(rp) >= 0)
        assert np.array_equal(np.diff(rp), np.bincount(src, minlength=int(lat.n_rows[b])))


def test_dense_and_arc_front_ends_agree():
    lats = [synth.layered_lattice(s, n_states=120, avg_degree=6.0, vocab=64, width=6, span=4) for s in (1, 2, 3)]
    a = LatticeBatch.from_synth(lats)
    em, tr = synth.collate_dense([l.dense() for l in lats])
    d = LatticeBatch.from_dense(em, tr)
    for k in ("arc_src", "arc_dst", "arc_label", "fwd_stream", "bwd_stream", "fwd_perm", "bwd_perm"):
        assert torch.equal(getattr(a, k), getattr(d, k)), k


@pytest.mark.parametrize("opts", [dict(), dict(slots_per_lane=1), dict(slots_per_lane=2), dict(slots_per_lane=4)])
def test_streams_replay_to_oracle_values(opts):
    lats = _cases()
    theta = synth.label_scores(9, 64)
    for l in lats:
        lat = LatticeBatch.from_synth([l], **opts)
        extra = lat.arc_w.numpy() if l.weight is not None else None
        sc = theta[l.label].astype(np.float64) + (l.weight.astype(np.float64) if l.weight is not None else 0.0)
        r = O.forward_backward(l.n_rows, l.src, l.dst, sc)
        beta = replay(lat, 0, "bwd", theta, extra)
        alpha = replay(lat, 0, "fwd", theta, extra)
        assert np.allclose(beta, r["logbeta"], atol=1e-9, equal_nan=True)
        assert np.allclose(alpha, r["logalpha"], atol=1e-9, equal_nan=True)
        assert lat.depth[0] >= 2 and lat.sink[0] == l.n_rows - 1


def test_huge_degree_state_is_split_into_continuation_pieces():
    # star: 0 -bos-> 1, 1 -> {2..201} (200 arcs), all -> 202 (in-degree 200), 202 -eos-> sink 203
    V = 256
    src = [0] + [1] * 200 + list(range(2, 202)) + [202]
    lab = [synth.BOS] + list(range(3, 203)) + [5] * 200 + [synth.EOS]
    dst = [1] + list(range(2, 202)) + [202] * 200 + [203]
    l = synth._finish(204, V, src, lab, dst)
    theta = synth.label_scores(2, V)
    r = O.forward_backward(l.n_rows, l.src, l.dst, theta[l.label].astype(np.float64))
    for U in (1, 2, 4):
        for mode in (0, 1, 2):  # packer's choice, narrow groups (tree of partial groups + combine), wide groups
            for nc in (False, True):  # compact (24-bit records, U = 4 only) and 32-bit records
                lat = LatticeBatch.from_synth([l], slots_per_lane=U, group_mode=mode, no_compact=nc)
                assert np.allclose(replay(lat, 0, "bwd", theta), r["logbeta"], atol=1e-9)
                assert np.allclose(replay(lat, 0, "fwd", theta), r["logalpha"], atol=1e-9)
            if mode == 1:  # 200 arcs > 2 x 8 lanes x U: scratch rows are in use
                assert lat.max_rows > l.n_rows
    lat = LatticeBatch.from_synth([l], slots_per_lane=1)  # 200 in-arcs > 64 lanes x 1 slot
    m = lat.meta_host[0]
    s = lat.fwd_stream.numpy().view(np.uint32)[int(m[_lib.META_FWD_OFF]):][: int(m[_lib.META_FWD_TILES]) * 128]
    ctl = s.reshape(-1, 128)[:, :64]
    assert np.any((ctl >> 30) & 1)  # continuation pieces (carry record + flag) are used


def test_pack_rejects_bad_lattices():
    V = 8
    def arcs(src, lab, dst, n):
        return dict(n_rows=np.array([n], np.int32), arc_off=np.array([0, len(src)], np.int64), src=np.array(src, np.int32),
                    label=np.array(lab, np.int32), dst=np.array(dst, np.int32), vocab=V)
    with pytest.raises(_lib.NfstError) as e:  # 0 -> 1 -> 2 -> 1
        LatticeBatch.from_arcs(**arcs([0, 1, 2, 2], [1, 3, 3, 4], [1, 2, 1, 3], 4))
    assert e.value.code == -3 and e.value.lattice == 0
    with pytest.raises(_lib.NfstError) as e:  # two states without out arcs
        LatticeBatch.from_arcs(**arcs([0, 0], [3, 4], [1, 2], 3))
    assert e.value.code == -4
    with pytest.raises(_lib.NfstError) as e:  # same (state, label) twice
        LatticeBatch.from_arcs(**arcs([0, 0, 1], [3, 3, 2], [1, 1, 2], 3))
    assert e.value.code == -5
    with pytest.raises(_lib.NfstError) as e:  # state index out of range
        LatticeBatch.from_arcs(**arcs([0], [3], [5], 2))
    assert e.value.code == -2
    with pytest.raises(_lib.NfstError) as e:  # more rows than the LDS-resident engine takes
        LatticeBatch.from_arcs(**arcs([0], [3], [1], 9000))
    assert e.value.code == -6


def test_trivial_and_padded_lattices():
    V = 8
    # a single arc 0 -eos-> sink, padded with junk rows that claim arcs everywhere (pad id 3)
    l = synth._finish(2, V, [0], [synth.EOS], [1])
    em, tr = l.dense()
    em, tr = em[None], tr[None]
    big = np.full((1, 6, V), 3, dtype=tr.dtype); big[0, :2] = tr[0]
    bem = np.ones((1, 6, V), dtype=bool); bem[0, :2] = em[0]
    lat = LatticeBatch.from_dense(bem, big)
    assert lat.total_arcs == 2 and lat.total_dp_arcs == 1 and lat.sink[0] == 1 and lat.depth[0] == 1
    assert lat.n_rows[0] == 6


def test_synth_is_deterministic():
    a = synth.layered_lattice(1234)
    b = synth.layered_lattice(1234)
    assert np.array_equal(a.src, b.src) and np.array_equal(a.label, b.label) and np.array_equal(a.dst, b.dst)
    assert 1800 <= a.n_rows <= 2202 and 15000 < a.n_arcs < 26000
    # the checksum pins the generator across machines / numpy builds
    h = int((a.src.astype(np.int64) * 31 + a.label * 17 + a.dst).sum() % (2 ** 31))
    assert h == int((b.src.astype(np.int64) * 31 + b.label * 17 + b.dst).sum() % (2 ** 31))


def test_concat_of_packed_parts_equals_packing_the_batch():
    """Pack once per example, batch by concatenation: the result is bit-identical to
    packing the whole batch, for plain and weighted lattices and for parts of several lattices."""
    import torch
    for weighted in (False, True):
        lats = [synth.layered_lattice(70 + i, n_states=40 + 31 * i, avg_degree=5.0, vocab=64, width=1 + i % 4, span=3, weighted=weighted)
                for i in range(6)]
        whole = LatticeBatch.from_synth(lats)
        for groups in ([[l] for l in lats], [lats[:2], lats[2:3], lats[3:]]):
            cat = LatticeBatch.concat([LatticeBatch.from_synth(g) for g in groups])
            for k in LatticeBatch._FIELDS:
                a, b = getattr(whole, k), getattr(cat, k)
                assert (a is None and b is None) or (a.dtype == b.dtype and torch.equal(a, b)), k
            assert whole._h == cat._h
    with pytest.raises(ValueError):
        LatticeBatch.concat([LatticeBatch.from_synth(lats[:1]),
                             LatticeBatch.from_synth([synth.layered_lattice(1, n_states=30, avg_degree=4.0, vocab=32, width=2, span=2)])])


def test_packed_sidecar_round_trip(tmp_path):
    """8f-1: an example's .npz is packed once, kept beside it, and a batch is the concatenation of
    loaded sidecars -- bit-identical to packing the collated dense tables."""
    from nfst_amd import io
    V, pad = 24, 0
    lats = [synth.layered_lattice(90 + i, n_states=18 + 9 * i, avg_degree=3.0, vocab=V, width=2, span=2) for i in range(3)]
    dense = [l.dense() for l in lats]
    names = []
    for i, (em, tr) in enumerate(dense):
        f = str(tmp_path / f"ex{i}.npz")
        io.save_fsa_npz(f, (em, tr), (em, tr), np.arange(3), np.arange(4))
        names.append(f)
    first = [io.load_packed(f) for f in names]
    assert all(os.path.exists(io.packed_sidecar(f)) for f in names)
    again = [io.load_packed(f) for f in names]  # from the sidecars
    for a, b in zip(first, again):
        assert a._h == b._h
        for k in LatticeBatch._FIELDS:
            x, y = getattr(a, k), getattr(b, k)
            assert (x is None and y is None) or (x.dtype == y.dtype and torch.equal(x, y)), k
    batch = io.collate_packed(again)
    col = io.collate([io.load_fsa_from_npz(f) for f in names], pad)
    whole = LatticeBatch.from_dense(col[0], col[1])
    assert np.array_equal(batch.n_arcs, whole.n_arcs)
    for k in ("arc_src", "arc_dst", "arc_label", "fwd_stream", "bwd_stream", "fwd_perm", "bwd_perm"):
        assert torch.equal(getattr(batch, k), getattr(whole, k)), k
    # a file that is not a sidecar, and a sidecar of another ABI version, are refused
    with pytest.raises(ValueError):
        LatticeBatch.load(names[0])
    side = io.packed_sidecar(names[0])
    raw = bytearray(open(side, "rb").read())
    stale = bytearray(raw)
    stale[8:16] = np.int64(1).tobytes()  # the ABI word
    open(side, "wb").write(stale)
    with pytest.raises(ValueError, match="ABI"):
        LatticeBatch.load(side)
    os.utime(side, (os.path.getmtime(names[0]) + 10,) * 2)
    assert io.load_packed(names[0])._h == first[0]._h  # rebuilt


def test_sidecar_content_checks(tmp_path):
    """A damaged, truncated or inconsistent sidecar never reaches a kernel: every array carries a CRC-32C
    (``nfst_crc32c``) that ``load`` verifies, and ``nfst_validate_batch`` checks every offset, count and id the
    kernels turn into addresses -- with the checksums switched off too."""
    from nfst_amd import _lib
    lats = [synth.layered_lattice(50 + i, n_states=60 + 20 * i, avg_degree=4.0, vocab=40, width=3, span=2, weighted=(i == 1)) for i in range(2)]
    for l in lats:
        lat = LatticeBatch.from_synth([l])
        lat.validate()
        f = str(tmp_path / "x.nfstpk")
        lat.save(f)
        back = LatticeBatch.load(f)
        assert back._h == lat._h
        for k in LatticeBatch._FIELDS:
            x, y = getattr(lat, k), getattr(back, k)
            assert (x is None and y is None) or (x.dtype == y.dtype and torch.equal(x, y)), k
        raw = bytearray(open(f, "rb").read())
        # one flipped byte in the middle of the arrays: the checksum catches it
        bad = bytearray(raw)
        bad[len(bad) // 2] ^= 0x40
        open(f, "wb").write(bad)
        with pytest.raises(ValueError, match="checksum"):
            LatticeBatch.load(f)
        # truncated file
        open(f, "wb").write(raw[: len(raw) - 4096])
        with pytest.raises(ValueError):
            LatticeBatch.load(f)
        open(f, "wb").write(raw)
        # structural damage with the checksums off: a tile record that names a state beyond max_rows, a slot -> arc
        # map entry beyond the lattice's arcs, a meta record whose program runs past the stream
        for field, poison in (("bwd_stream", None), ("fwd_perm", 1 << 28), ("meta", None), ("arc_dst", 1 << 20), ("row_ptr", -5)):
            t = {k: (None if v is None else v.clone()) for k, v in lat._t.items()}
            if field == "bwd_stream":
                t[field][1] = t[field][1] | 0x1fff  # first record of the first lane: state 8191
            elif field == "meta":
                t[field][_lib.META_BWD_TILES] += 100000
            else:
                t[field][3] = poison
            broken = LatticeBatch(lat._h, t)
            with pytest.raises(_lib.NfstError):
                broken.validate()
            broken.save(f)
            with pytest.raises(ValueError):
                LatticeBatch.load(f, verify=False)
    # the checksum itself: CRC-32C of "123456789" is 0xE3069283; chaining through the seed
    msg = np.frombuffer(b"123456789", dtype=np.uint8)
    assert _lib.lib.nfst_crc32c(msg.ctypes.data, 9, 0) == 0xE3069283
    part = _lib.lib.nfst_crc32c(msg.ctypes.data, 4, 0)
    assert _lib.lib.nfst_crc32c(msg[4:].ctypes.data, 5, part) == 0xE3069283


def test_concat_into_arena_and_threads():
    """``concat`` into the grow-only buffers of a ``HostArena`` (reused from batch to batch) and with several host
    threads gives the arrays of the plain concatenation."""
    from nfst_amd.lattice import HostArena
    lats = [synth.layered_lattice(70 + i, n_states=40 + 31 * i, avg_degree=5.0, vocab=64, width=1 + i % 4, span=3) for i in range(7)]
    parts = [LatticeBatch.from_synth([l]) for l in lats]
    arena = HostArena(pin=False)
    for sel in (parts, parts[:3], parts[2:]):  # the arena's buffers are reused and outgrown
        ref = LatticeBatch.concat(sel, n_threads=1)
        got = LatticeBatch.concat(sel, arena=arena, n_threads=4)
        assert got._h == ref._h
        for k in LatticeBatch._FIELDS:
            x, y = getattr(ref, k), getattr(got, k)
            assert (x is None and y is None) or torch.equal(x, y), k
        got.validate()


def test_graft_entry_build_runs():
    """The driver's build check: compiles (or finds up to date) the HIP library and the oracle and
    verifies the library against the header's ABI version."""
    import __graft_entry__ as g
    g.build()


def _random_dag(rng, n_inner, vocab, hub):
    """A random deterministic acyclic lattice: state 0 -bos-> 1, inner states 1 .. n_inner in
    topological order with random fan-out to later states (distinct labels per state, parallel
    arcs between a state pair allowed), optionally one hub that every earlier state feeds and that
    feeds every later one, last inner state -eos-> sink."""
    last, sink = n_inner, n_inner + 1
    src, lab, dst = [0], [synth.BOS], [1]
    for s in range(1, last):
        hi = last
        deg = int(min(rng.integers(1, 9), vocab - 3))
        targets = rng.integers(s + 1, hi + 1, size=deg)
        if hub and s < hub:
            targets[0] = hub
        if s == hub:
            targets = np.arange(s + 1, min(hi, s + vocab - 3) + 1)
        labels = rng.choice(np.arange(3, vocab), size=len(targets), replace=False)
        for t, l in zip(targets, labels):
            src.append(s); lab.append(int(l)); dst.append(int(t))
    src.append(last); lab.append(synth.EOS); dst.append(sink)
    # drop states nothing reaches (their arcs would dangle): keep arcs whose source is reachable
    reach = {0}
    for s, d in sorted(zip(src, dst)):
        if s in reach:
            reach.add(d)
    keep = [i for i, s in enumerate(src) if s in reach]
    return synth._finish(sink + 1, vocab, [src[i] for i in keep], [lab[i] for i in keep], [dst[i] for i in keep])


def test_random_dags_under_every_packing():
    """Property test (hypothesis): whatever the shape -- chains, hubs with a fan-in and fan-out of
    dozens, parallel arcs -- every packing replays to the oracle's alpha and beta and keeps the
    schedule invariants stream_check asserts."""
    from hypothesis import given, settings, strategies as st, HealthCheck

    @settings(max_examples=40, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
    @given(seed=st.integers(0, 10 ** 6), n_inner=st.integers(2, 90), hub_on=st.booleans(),
           U=st.sampled_from([0, 1, 2, 4]), mode=st.sampled_from([0, 1, 2]), nc=st.booleans())
    def check(seed, n_inner, hub_on, U, mode, nc):
        ran.append(0)
        rng = np.random.default_rng(seed)
        V = 96
        l = _random_dag(rng, n_inner, V, hub=int(rng.integers(2, n_inner)) if hub_on and n_inner > 3 else 0)
        theta = synth.label_scores(seed % 97, V)
        try:
            r = O.forward_backward(l.n_rows, l.src, l.dst, theta[l.label].astype(np.float64))
        except O.OracleError:
            return  # not a lattice (a dead end beside the sink): the packer refuses it as well
        lat = LatticeBatch.from_synth([l], slots_per_lane=U, group_mode=mode, no_compact=nc)
        assert np.allclose(replay(lat, 0, "bwd", theta), r["logbeta"], atol=1e-9, equal_nan=True)
        assert np.allclose(replay(lat, 0, "fwd", theta), r["logalpha"], atol=1e-9, equal_nan=True)
        ran[-1] = l.n_arcs

    ran = []
    check()
    assert sum(1 for x in ran if x > 0) >= 30 and max(ran) > 150


def test_device_pack_layout_matches_host_packer():
    """``nfst_pack_device_layout`` (the host half of the device packer: counts -> offsets, sizes, flags) reproduces the
    host packer's meta records and header from the host packer's own counts."""
    import ctypes as C
    for lats in ([synth.layered_lattice(70 + i, n_states=40 + 31 * i, avg_degree=5.0, vocab=64, width=1 + i % 4, span=3) for i in range(6)],
                 [synth.layered_lattice(80 + i, n_states=120, avg_degree=6.0, vocab=48, width=5, span=2, weighted=True) for i in range(3)],
                 synth.bench_batch(2)):
        for gm in (0, 1, 2):
            ref = LatticeBatch.from_synth(lats, group_mode=gm)
            B = ref.n_lattices
            meta = ref.meta_host.copy()
            for col in (_lib.META_ROW_OFF, _lib.META_ARC_OFF, _lib.META_FWD_OFF, _lib.META_BWD_OFF, _lib.META_FWD_SLOT_OFF, _lib.META_BWD_SLOT_OFF):
                meta[:, col] = -1  # what the planning kernel leaves to the host
            meta = np.ascontiguousarray(meta.reshape(-1))
            status = np.zeros(B, np.int32)
            scratch = np.full(B, ref.max_rows - int(ref.n_rows.max()), np.int32) if len(set(ref.n_rows)) == 1 else None
            if scratch is None:  # scratch rows per lattice are not kept by the host packer: any split with the same maximum will do
                scratch = np.zeros(B, np.int32)
                scratch[int(np.argmax(ref.n_rows))] = ref.max_rows - int(ref.n_rows.max())
            header = _lib.Batch()
            bad = C.c_int32(-1)
            rc = _lib.lib.nfst_pack_device_layout(meta.ctypes.data, status.ctypes.data, scratch.ctypes.data, B, ref.vocab, ref.weighted, C.byref(header), C.byref(bad))
            assert rc == 0
            assert np.array_equal(meta.reshape(B, -1), ref.meta_host)
            for k in LatticeBatch._HEADER:
                if k == "max_rows" and header.max_rows != ref._h[k]:
                    assert header.max_rows >= int(ref.n_rows.max())  # (the split of the scratch rows above is a guess)
                    continue
                assert int(getattr(header, k)) == ref._h[k], k
    # a failed lattice is reported with its code
    status = np.array([0, -3, 0], np.int32)
    rc = _lib.lib.nfst_pack_device_layout(np.zeros(48, np.int32).ctypes.data, status.ctypes.data, np.zeros(3, np.int32).ctypes.data, 3, 8, 0,
                                          C.byref(_lib.Batch()), C.byref(bad))
    assert rc == -3 and bad.value == 1


def test_tuning_switches_are_checked():
    assert _lib.lib.nfst_tuning_set(b"no such switch", 1) == -1
    with _lib.tuning(tw=0, precise=1, neu_bf16=0):
        pass
    with pytest.raises(ValueError):
        _lib.tuning(nonsense=1)


def test_build_guard_on_register_reports():
    """``nfst_amd.build.check_resources``: the build fails when a fused sweep does not own exactly the 32 AGPRs it stages tiles in,
    when a sweep kernel spills vector registers, or when a kernel that stages nothing in AGPRs uses some (the round-2 abort:
    DESIGN.md section 4.1); the report of the library that is loaded passes."""
    import json
    from nfst_amd import build
    ok = {"k_forward_backward<512, 0, true, false, false>": dict(vgprs=62, agprs=32, vgpr_spill=0),
          "k_forward_backward<1024, 0, false, true, false>": dict(vgprs=60, agprs=0, vgpr_spill=0),
          "k_backward<512, 0, true, false>": dict(vgprs=67, agprs=0, vgpr_spill=0),
          "k_backward_neural_grad<4>": dict(vgprs=128, agprs=0, vgpr_spill=9)}  # (not a sweep kernel: its spills are its own business)
    assert build.check_resources(ok) == []
    for name, patch in (("k_forward_backward<512, 0, true, false, false>", dict(agprs=40)),   # the compiler took AGPRs for itself
                        ("k_forward_backward<512, 0, true, false, false>", dict(agprs=0)),
                        ("k_forward_backward<512, 0, true, false, false>", dict(vgpr_spill=2)),
                        ("k_forward_backward<1024, 0, false, true, false>", dict(vgpr_spill=1)),
                        ("k_backward<512, 0, true, false>", dict(agprs=4))):
        bad = {k: dict(v) for k, v in ok.items()}
        bad[name].update(patch)
        assert build.check_resources(bad), (name, patch)
    # the remarks parser and the in-house demangler
    text = ("x.h:1:1: remark: Function Name: _ZN12_GLOBAL__N_118k_forward_backwardILi256ELi0ELb1ELb0ELb0EEEv10nfst_batch [-Rpass-analysis=kernel-resource-usage]\n"
            "x.h:1:1: remark:     VGPRs: 60 [-Rpass-analysis=kernel-resource-usage]\n"
            "x.h:1:1: remark:     AGPRs: 32 [-Rpass-analysis=kernel-resource-usage]\n"
            "x.h:1:1: remark:     VGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]\n")
    res = build.parse_resources(text)
    names = build._demangle(list(res))
    assert list(names.values()) == ["k_forward_backward<256, 0, true, false, false>"]
    assert list(res.values())[0] == dict(vgprs=60, agprs=32, vgpr_spill=0)
    rep = os.path.join(os.path.dirname(build.OUT), "libnfst_hip.resources.json")
    if os.path.exists(rep):
        assert build.check_resources(json.load(open(rep))) == []


def test_struct_sizes_of_the_bindings_match_the_library():
    """the entry points copy whole structs: nfst_amd/_lib.py refuses to import when one of its ctypes declarations has drifted
    from the header (nfst_sizeof); the stub of INTEGRATION.md declares the same nfst_batch"""
    import ctypes as C
    import os
    import re
    for name, cls in (("nfst_batch", _lib.Batch), ("nfst_scores", _lib.Scores), ("nfst_chunks", _lib.Chunks), ("nfst_chunk_opts", _lib.ChunkOpts),
                      ("nfst_pack_opts", _lib.PackOpts), ("nfst_step_extras", _lib.StepExtras), ("nfst_arcs_device", _lib.ArcsDevice)):
        assert _lib.lib.nfst_sizeof(name.encode()) == C.sizeof(cls), name
    assert _lib.lib.nfst_sizeof(b"no_such_struct") == -1
    doc = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    stub = doc[doc.index("class Batch(C.Structure)"):doc.index("class Scores(C.Structure)")]
    n32 = len(re.findall(r'"(n_lattices|vocab|max_rows|max_tiles|weighted|reserved0|only_tag|reserved2)"', stub))
    n64 = len(re.findall(r'"(total_rows|total_arcs|total_dp_arcs|fwd_words|bwd_words|fwd_slots|bwd_slots)"', stub))
    nptr = len(re.findall(r'"(meta|row_ptr|arc_src|arc_dst|arc_label|arc_w|fwd_stream|bwd_stream|fwd_perm|bwd_perm|arc_sd|arc_l16|chunks|only)"', stub))
    assert 4 * n32 + 8 * n64 + 8 * nptr == C.sizeof(_lib.Batch)
