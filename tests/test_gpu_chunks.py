"""The chunked flavour of the sweeps for deep, narrow lattices (chunk_kernels.h) against the oracle and against the general
kernels, through the C ABI (``-m gpu``).  Host-side checks of the programs: tests/test_chunks_cpu.py."""
import os

import numpy as np
import pytest
import torch

from nfst_amd import _lib, ops, synth
from nfst_amd.lattice import LatticeBatch
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _oracle(l, theta, asc=None):
    sc = theta[l.label].astype(np.float64)
    if l.weight is not None:
        sc = sc + l.weight.astype(np.float64)
    if asc is not None:
        sc = sc + asc.astype(np.float64)
    return O.forward_backward(l.n_rows, l.src, l.dst, sc)


def _canonical(l):
    return np.lexsort((l.label, l.src))


def _check(lat, lats, r, theta, asc=None, tol_z=1e-8, thetas=None, loose=()):
    la, lb, post = r.logalpha.cpu().numpy(), r.logbeta.cpu().numpy(), r.posterior.cpu().numpy()
    worst = 0.0
    for b, l in enumerate(lats):
        a0, na, r0, nr = int(lat.arc_off[b]), int(lat.n_arcs[b]), int(lat.row_off[b]), int(lat.n_rows[b])
        order = _canonical(l)
        o = _oracle(l, theta if thetas is None else thetas[b], None if asc is None else asc[a0:a0 + na][np.argsort(order)])
        if b not in loose:
            worst = max(worst, abs(float(r.logz64[b]) - o["logZ"]))
        assert abs(float(r.logz64[b]) - o["logZ"]) <= (2e-5 if b in loose else tol_z), b
        assert abs(float(r.logz[b]) - o["logZ"]) <= 1e-6 * max(1.0, abs(o["logZ"]))
        for got, ref in ((la[r0:r0 + nr], o["logalpha"]), (lb[r0:r0 + nr], o["logbeta"])):
            fin = np.isfinite(ref)
            assert np.array_equal(np.isfinite(got), fin), b
            assert np.all(np.abs(got[fin] - ref[fin]) <= 1e-6 * np.maximum(1.0, np.abs(ref[fin]))), b
        assert np.max(np.abs(post[a0:a0 + na] - o["posterior"][order])) <= 2e-6, b
    return worst


def test_snips_shaped_batch_runs_the_chunked_flavour(dev):
    """BASELINE configs[2]: the packer cuts chunked programs on the way to the device; every output against the oracle, and
    against the general kernels on the same batch"""
    V = 250
    lats = synth.snips_shaped_batch(64, vocab=V)
    theta = synth.label_scores(64, V, mean=-1.5, std=0.8)
    lat = LatticeBatch.from_synth(lats, device=dev)
    assert lat.chunks is not None and lat.chunks.ws is not None
    th = torch.from_numpy(theta)
    r = ops.forward_backward(lat, th, want_grad_theta=True)
    assert not lat.chunks.flagged().any()
    worst = _check(lat, lats, r, theta)
    assert worst <= 1e-9  # float64 throughout
    with _lib.tuning(chunked=0):
        g = ops.forward_backward(lat, th, want_grad_theta=True)
    assert torch.max(torch.abs(r.logz64 - g.logz64)).item() <= 1e-8
    assert torch.max(torch.abs(r.posterior - g.posterior)).item() <= 2e-6
    assert torch.max(torch.abs(r.grad_theta - g.grad_theta)).item() <= 1e-4
    assert torch.allclose(r.logbeta, g.logbeta, rtol=1e-6, atol=1e-6, equal_nan=True)
    b = ops.backward(lat, th)
    assert torch.equal(b.logz64, r.logz64) and torch.equal(b.logbeta, r.logbeta)
    again = ops.forward_backward(lat, th, want_grad_theta=True)
    for name, x, y in zip(r._fields, r, again):
        if name == "grad_theta":  # (float atomics in LDS: the order of the additions is not fixed)
            assert torch.allclose(x, y, rtol=1e-5, atol=1e-6)
        else:
            assert (x is None and y is None) or torch.equal(x, y), name


@pytest.mark.parametrize("opts", [dict(), dict(threads=64), dict(threads=256, max_chunks=3), dict(max_chunks=1), dict(threads=128, lds_bytes=8192)])
def test_forced_programs_on_small_lattices_with_every_kind_of_score(dev, opts):
    """chains, single-level lattices, odd frontier sizes; per-lattice theta, table weights and caller scores"""
    shapes = [(60, 3.0, 2, 1), (90, 4.0, 3, 2), (40, 2.0, 1, 1), (120, 5.0, 4, 2), (30, 3.0, 2, 3), (7, 2.0, 2, 1), (200, 3.0, 6, 1),
              (4, 1.0, 1, 1), (400, 2.5, 5, 2)]
    lats = [synth.layered_lattice(100 + i, n_states=n, avg_degree=deg, vocab=40, width=w, span=sp, max_degree=12, weighted=True)
            for i, (n, deg, w, sp) in enumerate(shapes)]
    host = LatticeBatch.from_synth(lats)
    assert host.build_chunks(force=True, **opts)
    lat = host.to(dev)
    rng = np.random.default_rng(5)
    thetas = np.stack([synth.label_scores(20 + b, 40, mean=-0.5, std=0.7) for b in range(len(lats))])
    asc = (0.3 * rng.standard_normal(lat.total_arcs)).astype(np.float32)
    r = ops.forward_backward(lat, torch.from_numpy(thetas), arc_scores=torch.from_numpy(asc))
    assert not lat.chunks.flagged().any()
    _check(lat, lats, r, None, asc=asc, thetas=thetas)
    b = ops.backward(lat, torch.from_numpy(thetas), arc_scores=torch.from_numpy(asc))
    assert torch.equal(b.logz64, r.logz64)


def test_lattices_beyond_the_range_go_back_to_the_general_kernels(dev):
    """pass 1 computes in plain float64: a weight or a partial sum beyond 2^+-480 flags its lattice on the device and the
    general kernels run it in the same call; masked arcs (-inf) and ordinary scores stay on the chunked path"""
    V = 40
    lats = [synth.layered_lattice(300 + i, n_states=150, avg_degree=3.0, vocab=V, width=3, span=1, max_degree=10) for i in range(6)]
    thetas = np.stack([synth.label_scores(b, V, mean=-1.0, std=0.5) for b in range(6)])
    thetas[1, 5] = -400.0           # one weight below 2^-480
    thetas[2, :] -= 20.0            # sums shrink by e^-20 per level: beyond the range within a chunk of 25 levels
    thetas[3, 7] = -np.inf          # a masked label: exact zeros, no flag
    thetas[4, :] += 16.0            # growing sums
    host = LatticeBatch.from_synth(lats)
    assert host.build_chunks(force=True, max_chunks=2)
    lat = host.to(dev)
    total = torch.zeros(3, dtype=torch.float64, device=dev)
    r = ops.forward_backward(lat, torch.from_numpy(thetas), want_grad_theta=True, total=total, total_slot=1)
    fl = lat.chunks.flagged()
    assert fl[1] and fl[2] and fl[4] and not fl[0] and not fl[3] and not fl[5]
    _check(lat, lats, r, None, thetas=thetas, loose=(1, 2, 4))  # (the flagged ones: the general float32 flavour's tolerance)
    assert abs(float(total[1]) - float(r.logz64.sum())) <= 1e-9 * abs(float(r.logz64.sum()))
    assert float(total[2]) == 0.0
    with _lib.tuning(chunked=0):
        g = ops.forward_backward(lat, torch.from_numpy(thetas), want_grad_theta=True)
    assert torch.equal(r.logz64[fl], g.logz64[fl])  # (the same kernels ran them)
    assert torch.max(torch.abs(r.grad_theta - g.grad_theta)).item() <= 1e-4
    b = ops.backward(lat, torch.from_numpy(thetas))
    assert torch.max(torch.abs(b.logz64 - r.logz64)).item() <= 1e-8


def test_log_z_autograd_through_the_chunked_flavour(dev):
    V = 64
    lats = synth.snips_shaped_batch(8, vocab=V, first_seed=4100)
    lat = LatticeBatch.from_synth(lats, device=dev)
    assert lat.chunks is not None
    theta = torch.from_numpy(synth.label_scores(3, V, mean=-1.0, std=0.6)).to(dev).requires_grad_()
    asc = torch.zeros(lat.total_arcs, device=dev, requires_grad=True)
    z = ops.log_z(lat, theta, asc)
    z.sum().backward()
    with _lib.tuning(chunked=0):
        theta2 = theta.detach().clone().requires_grad_()
        asc2 = torch.zeros(lat.total_arcs, device=dev, requires_grad=True)
        ops.log_z(lat, theta2, asc2).sum().backward()
    assert torch.max(torch.abs(asc.grad - asc2.grad)).item() <= 2e-6
    assert torch.max(torch.abs(theta.grad - theta2.grad)).item() <= 1e-3 * max(1.0, float(theta2.grad.abs().max()))


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("NFST_CHUNK_FUZZ_SEEDS", "16")))))  # (more seeds for a one-off soak)
def test_fuzz_chunked_flavour_against_oracle(dev, seed):
    """random narrow lattices (chains to eight states per level, arcs up to three levels ahead), random cuts (workgroup size, LDS
    budget, chunk limit), per-lattice theta, table weights and caller scores: log Z within 1e-8 of the float64 oracle, posteriors
    within 2e-6, for every lattice the chunked kernels kept"""
    rng = np.random.default_rng(9000 + seed)
    B = int(rng.integers(1, 20))
    V = int(rng.choice([24, 64, 300]))
    weighted = bool(rng.integers(0, 2))
    lats = []
    while len(lats) < B:
        try:
            lats.append(synth.layered_lattice(int(rng.integers(1, 1 << 30)), n_states=int(rng.choice([4, 6, 11, 40, 150, 600, 1500])),
                                              avg_degree=float(rng.choice([1.5, 3.0, 5.0])), vocab=V, width=int(rng.choice([1, 2, 3, 4, 6, 8])),
                                              span=int(rng.choice([1, 2, 3])), max_degree=min(10, (V - 12) // 2), weighted=weighted))
        except AssertionError:
            continue
    host = LatticeBatch.from_synth(lats)
    opts = dict(threads=int(rng.choice([64, 256, 512, 1024])), max_chunks=int(rng.choice([0, 0, 1, 2, 7])),
                lds_bytes=int(rng.choice([0, 16384, 65536])))
    if not host.build_chunks(force=True, **opts):
        pytest.skip("no cut within these limits")
    lat = host.to(dev)
    thetas = np.stack([synth.label_scores(int(rng.integers(1 << 20)), V, mean=float(rng.choice([-2.0, 0.0, 1.0])), std=float(rng.choice([0.3, 1.0])))
                       for _ in range(B)])
    asc = (0.5 * rng.standard_normal(lat.total_arcs)).astype(np.float32) if rng.integers(0, 2) else None
    r = ops.forward_backward(lat, torch.from_numpy(thetas), arc_scores=None if asc is None else torch.from_numpy(asc))
    fl = lat.chunks.flagged()
    _check(lat, lats, r, None, asc=asc, thetas=thetas, loose=tuple(np.flatnonzero(fl).tolist()))
    b = ops.backward(lat, torch.from_numpy(thetas), arc_scores=None if asc is None else torch.from_numpy(asc))
    assert torch.max(torch.abs(b.logz64 - r.logz64)).item() <= 2e-5


def test_chunked_programs_for_a_batch_packed_on_the_device(dev):
    """build_chunks on a device batch cuts the programs from a host copy of its canonical arrays"""
    lats = synth.snips_shaped_batch(6, vocab=64, first_seed=5200)
    n_rows, arc_off, src, label, dst, w = synth.batch_arcs(lats)
    lat = LatticeBatch.from_arcs_device(n_rows, arc_off, src, label, dst, 64, device=dev)
    assert lat.chunks is None
    theta = torch.from_numpy(synth.label_scores(2, 64, mean=-1.0, std=0.5))
    g = ops.forward_backward(lat, theta)
    assert lat.build_chunks(force=True) and lat.chunks.ws is not None
    r = ops.forward_backward(lat, theta)
    assert not lat.chunks.flagged().any()
    assert torch.max(torch.abs(r.logz64 - g.logz64)).item() <= 1e-7 and torch.max(torch.abs(r.posterior - g.posterior)).item() <= 2e-6
    _check(lat, lats, r, synth.label_scores(2, 64, mean=-1.0, std=0.5))


def test_more_lattices_than_cus_use_the_smaller_workgroups(dev):
    """2 B > 256 workgroups: the programs are cut for 512 threads and 64 KiB of LDS (two workgroups per CU)"""
    rng = np.random.default_rng(77)
    V = 48
    lats = [synth.layered_lattice(7000 + i, n_states=int(rng.integers(30, 160)), avg_degree=3.0, vocab=V, width=int(rng.choice([1, 2, 3])),
                                  span=int(rng.choice([1, 2])), max_degree=8) for i in range(160)]
    host = LatticeBatch.from_synth(lats)
    assert host.build_chunks(force=True)
    assert host.chunks._h["threads"] == 512 and host.chunks._h["lds_bytes"] <= 64 * 1024
    lat = host.to(dev)
    theta = synth.label_scores(5, V, mean=-0.8, std=0.6)
    r = ops.forward_backward(lat, torch.from_numpy(theta), want_grad_theta=True)
    assert not lat.chunks.flagged().any()
    _check(lat, lats, r, theta)
    with _lib.tuning(chunked=0):
        g = ops.forward_backward(lat, torch.from_numpy(theta), want_grad_theta=True)
    assert torch.max(torch.abs(r.grad_theta - g.grad_theta)).item() <= 1e-4


def test_beta_pairs_and_sampling_through_the_chunked_flavour(dev):
    """beta_me (float32 mantissa / exponent pairs, what nfst_sample_paths walks on) from the chunked sweeps: the same numbers as
    the general kernels', and posterior samples whose log q is path score - log Z"""
    V = 64
    lats = synth.snips_shaped_batch(6, vocab=V, first_seed=6100)
    host = LatticeBatch.from_synth(lats)
    assert host.build_chunks(force=True)
    lat = host.to(dev)
    theta = synth.label_scores(4, V, mean=-1.0, std=0.6)
    th = torch.from_numpy(theta)
    b = ops.backward(lat, th, want_me=True)
    assert not lat.chunks.flagged().any()
    with _lib.tuning(chunked=0):
        g = ops.backward(lat, th, want_me=True)
    mb, mg = b.beta_me.cpu().numpy(), g.beta_me.cpu().numpy()
    eb, eg = mb[:, 1].view(np.int32), mg[:, 1].view(np.int32)
    live = mg[:, 0] > 0
    assert np.array_equal(mb[:, 0] > 0, live)
    vb = np.log(mb[live, 0].astype(np.float64)) + eb[live] * np.log(2.0)
    vg = np.log(mg[live, 0].astype(np.float64)) + eg[live] * np.log(2.0)
    assert np.max(np.abs(vb - vg)) <= 2e-6
    K = 64
    s = ops.sample_paths(lat, th, K, seed=3, beta=b)
    arcs, lens, logq = s.arcs.cpu().numpy(), s.lengths.cpu().numpy(), s.logq.cpu().numpy()
    for i, l in enumerate(lats):
        order = _canonical(l)
        sc = theta[l.label].astype(np.float64)[order]
        for k in range(0, K, 7):
            a = arcs[i, k, :lens[i, k]] - int(lat.arc_off[i])
            assert abs(sc[a].sum() - float(b.logz64[i]) - logq[i, k]) <= 5e-5 * max(1.0, abs(float(b.logz64[i])) / 100)


def test_chunked_programs_through_the_device_prefetcher(dev):
    """io.DevicePrefetcher (pinned staging, copies on a side stream): batches of deep, narrow lattices arrive with their chunked
    programs cut and their scratch in place; results are those of a plain blocking copy"""
    from nfst_amd import io
    V = 64
    theta = torch.from_numpy(synth.label_scores(5, V, mean=-1.0, std=0.5))
    cpu = [LatticeBatch.from_synth([synth.layered_lattice(900 + 10 * i + j, n_states=400 + 100 * j, avg_degree=3.0, vocab=V, width=2, span=1,
                                                           max_degree=8) for j in range(3)]) for i in range(4)]
    want = [ops.forward_backward(b.to(dev), theta) for b in cpu]
    assert all(b.chunks is not None for b in cpu)  # (cut on the way to the device)
    got = []
    for b in io.DevicePrefetcher(cpu, dev):
        assert b.chunks is not None and b.chunks.ws is not None and b.chunks.ws.device.type == "cuda"
        got.append(ops.forward_backward(b, theta))
    assert len(got) == len(want)
    for a, w in zip(got, want):
        assert torch.equal(a.logz64, w.logz64) and torch.equal(a.posterior, w.posterior)
