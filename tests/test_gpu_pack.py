"""The packer on the device against the host packer (``-m gpu``, through the C ABI): bit for bit.

``nfst_pack_device_plan`` / ``_emit`` and ``nfst_dense_to_arcs_count`` / ``_write`` (include/nfst_hip.h) must give
exactly the arrays of ``nfst_pack_arcs`` / ``nfst_pack_dense`` -- canonical arcs, row pointers, tile programs,
slot -> arc maps, meta records, header -- on the corpus of tests/test_pack_cpu.py, on the shapes of the BASELINE
configs, under every group mode, and refuse what the host packer refuses with the same error codes."""
import numpy as np
import pytest
import torch

from nfst_amd import _lib, ops, synth
from nfst_amd.lattice import LatticeBatch

pytestmark = pytest.mark.gpu


def same(a: LatticeBatch, b: LatticeBatch, what=""):
    assert a._h == b._h, (what, a._h, b._h)
    assert np.array_equal(a.meta_host, b.meta_host), what
    for k in LatticeBatch._FIELDS:
        x, y = a._t[k], b._t[k]
        assert (x is None) == (y is None), (what, k)
        if x is not None:
            x, y = x.cpu(), y.cpu()
            if not torch.equal(x, y):
                bad = (x != y).nonzero().flatten()
                raise AssertionError(f"{what}: {k} differs at {bad.numel()} of {x.numel()} entries, first {int(bad[0])}: "
                                     f"{int(x[bad[0]])} vs {int(y[bad[0]])}")


def corpus():
    V = 256
    star_src = [0] + [1] * 200 + list(range(2, 202)) + [202]
    star_lab = [synth.BOS] + list(range(3, 203)) + [5] * 200 + [synth.EOS]
    star_dst = [1] + list(range(2, 202)) + [202] * 200 + [203]
    return {
        "small40": [synth.layered_lattice(3, n_states=30, avg_degree=3.0, vocab=40, width=4, span=2),
                    synth.edit_lattice([10, 11, 12, 13], [20, 21, 22], vocab=40, seed=2)],
        "mixed64": [synth.layered_lattice(4, n_states=200, avg_degree=8.0, vocab=64, width=9, span=5),
                    synth.layered_lattice(5, n_states=64, avg_degree=5.0, vocab=64, width=1, span=6),
                    synth.layered_lattice(7, n_states=120, avg_degree=6.0, vocab=64, width=6, span=4)],
        "weighted64": [synth.layered_lattice(6, n_states=150, avg_degree=6.0, vocab=64, width=7, span=3, weighted=True),
                       synth.layered_lattice(8, n_states=90, avg_degree=4.0, vocab=64, width=2, span=2, weighted=True)],
        "star200": [synth._finish(204, V, star_src, star_lab, star_dst)],  # fan-out and fan-in of 200: carry pieces, partial groups
        "trivial": [synth._finish(2, 8 + 3, [0], [synth.EOS], [1]), synth._finish(3, 8 + 3, [0, 1], [synth.BOS, synth.EOS], [1, 2])],
        "baseline": synth.bench_batch(6),
        "baseline_width4": synth.bench_batch(3, width=4),
        "snips": synth.snips_shaped_batch(5),
        "chains": [synth.layered_lattice(20 + i, n_states=n, avg_degree=2.0, vocab=24, width=1, span=1 + i % 2, max_degree=6, weighted=(i % 2 == 1)) for i, n in enumerate((4, 9, 17, 450))],
    }


@pytest.mark.parametrize("name", list(corpus()))
@pytest.mark.parametrize("group_mode", [0, 1, 2])
def test_device_packer_equals_host_packer(dev, name, group_mode):
    lats = corpus()[name]
    if name == "chains":  # (weighted and unweighted lattices do not share a batch)
        groups = [[l for l in lats if l.weight is None], [l for l in lats if l.weight is not None]]
    else:
        groups = [lats]
    for g in groups:
        n_rows, arc_off, src, label, dst, w = synth.batch_arcs(g)
        host = LatticeBatch.from_arcs(n_rows, arc_off, src, label, dst, g[0].vocab, arc_w=w, group_mode=group_mode)
        devb = LatticeBatch.from_arcs_device(n_rows, arc_off, src, label, dst, g[0].vocab, arc_w=w, device=dev, group_mode=group_mode)
        torch.cuda.synchronize()
        same(host, devb, f"{name} group_mode {group_mode}")
        devb.validate()
        # ... and the kernels run on it
        theta = torch.from_numpy(synth.label_scores(1, g[0].vocab))
        # (same kernels on both: a batch packed on the device has no chunked programs)
        a, b = ops.forward_backward(host.to(dev, auto_chunks=False), theta), ops.forward_backward(devb, theta)
        assert torch.equal(a.logz64, b.logz64) and torch.equal(a.posterior, b.posterior)


@pytest.mark.parametrize("pad", [0, 7])
@pytest.mark.parametrize("weighted", [False, True])
def test_dense_tables_on_the_device(dev, pad, weighted):
    """``set_masks`` with the collated tables already on the GPU (lightning.py:417): reachable rows -> arcs -> packed
    batch without a table leaving the device; pad-id padding rows (all-True rows, transition = pad) are ignored."""
    lats = [synth.layered_lattice(30 + i, n_states=40 + 50 * i, avg_degree=5.0, vocab=48, width=3 + i, span=3, weighted=weighted) for i in range(3)]
    lats.append(synth.edit_lattice([10, 11, 12], [20, 21, 22, 23], vocab=48, seed=3) if not weighted else
                synth.layered_lattice(39, n_states=25, avg_degree=3.0, vocab=48, width=2, span=2, weighted=True))
    em, tr = synth.collate_dense([l.dense(weighted=weighted) for l in lats], pad=pad)
    host = LatticeBatch.from_dense(em, tr)
    devb = LatticeBatch.from_dense(torch.from_numpy(em).to(dev), torch.from_numpy(tr).to(dev))
    torch.cuda.synchronize()
    assert devb.device.type == "cuda"
    same(host, devb, f"dense pad {pad} weighted {weighted}")


def test_device_packer_error_codes(dev):
    V = 8

    def arcs(src, lab, dst, n):
        return dict(n_rows=np.array([n], np.int32), arc_off=np.array([0, len(src)], np.int64), src=np.array(src, np.int32),
                    label=np.array(lab, np.int32), dst=np.array(dst, np.int32), vocab=V, device=dev)
    for bad, code in (((([0, 1, 2, 2], [1, 3, 3, 4], [1, 2, 1, 3], 4)), -3),   # 0 -> 1 -> 2 -> 1: cycle
                      ((([0, 0], [3, 4], [1, 2], 3)), -4),                      # two states without out arcs
                      ((([0, 0, 1], [3, 3, 2], [1, 1, 2], 3)), -5),             # same (state, label) twice
                      ((([0], [3], [5], 2)), -2),                               # state index out of range
                      ((([0], [3], [1], 9000)), -6)):                           # more rows than the engine takes
        with pytest.raises(_lib.NfstError) as e:
            LatticeBatch.from_arcs_device(**arcs(*bad))
        assert e.value.code == code, (bad, e.value.code)
        with pytest.raises(_lib.NfstError) as e2:
            LatticeBatch.from_arcs(**{k: v for k, v in arcs(*bad).items() if k != "device"})
        assert e2.value.code == code
    # a wide vocabulary is the host packer's: the device packer says so, from_dense falls back
    with pytest.raises(_lib.NfstError) as e:
        LatticeBatch.from_arcs_device(np.array([2], np.int32), np.array([0, 1], np.int64), np.array([0], np.int32), np.array([5], np.int32),
                                      np.array([1], np.int32), 3000, device=dev)
    assert e.value.code == -6


def test_baseline_batch_packed_on_the_device(dev):
    """BASELINE configs[1] (256 x ~2k states / ~20k arcs) from 12-byte arc lists: bit-identical to the host packer, and
    the step on it gives the same bits."""
    lats = synth.bench_batch(256)
    n_rows, arc_off, src, label, dst, w = synth.batch_arcs(lats)
    host = LatticeBatch.from_arcs(n_rows, arc_off, src, label, dst, 256)
    devb = LatticeBatch.from_arcs_device(n_rows, arc_off, src, label, dst, 256, device=dev)
    torch.cuda.synchronize()
    same(host, devb, "baseline 256")
