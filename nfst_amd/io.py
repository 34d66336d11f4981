"""On-disk lattice ingest: the reference's ``.npz`` records and its collate.

Keys ``num_emission, num_transition, denom_emission, denom_transition, gs, ps``
(/root/reference/src/preprocess/tr.py:182-190), loader
``Utils.load_fsa_from_npz`` (src/util/preprocess_util.py:293-323) and
``T9FSADataModule.collate`` (src/util/dataset_reader.py:175-186, padding with the
pad id via ``Utils.pad_sequence``, preprocess_util.py:368-392).  numpy only.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np

KEYS = ("num_emission", "num_transition", "denom_emission", "denom_transition", "gs", "ps")


def load_fsa_from_npz(npz_fname: str) -> Tuple[np.ndarray, ...]:
    with np.load(npz_fname, allow_pickle=False) as l:
        return tuple(l[k] for k in KEYS)


def save_fsa_npz(npz_fname: str, num, denom, gs, ps) -> None:
    np.savez_compressed(npz_fname, num_emission=num[0], num_transition=num[1], denom_emission=denom[0],
                        denom_transition=denom[1], gs=np.asarray(gs), ps=np.asarray(ps))


def pad_sequence(sequences: Sequence[np.ndarray], padding_value=0) -> np.ndarray:
    max_len = max(s.shape[0] for s in sequences)
    out = np.full((len(sequences), max_len) + sequences[0].shape[1:], fill_value=padding_value, dtype=sequences[0].dtype)
    for i, t in enumerate(sequences):
        out[i, : t.shape[0], ...] = t
    return out


def collate(batch: Sequence[Tuple[np.ndarray, ...]], pad: int):
    """Six padded arrays, like the reference's collate (dataset_reader.py:175-186)."""
    return tuple(pad_sequence([b[i] for b in batch], padding_value=pad) for i in range(6))


def packed_sidecar(npz_fname: str, which: str = "num") -> str:
    return npz_fname[: -len(".npz")] + f".{which}.nfst.npz" if npz_fname.endswith(".npz") else npz_fname + f".{which}.nfst.npz"


def load_packed(npz_fname: str, which: str = "num", cache: bool = True):
    """The ``num`` (or ``denom``) lattice of one example as a packed one-lattice batch.  The
    packer runs the first time an example is seen; its output is kept in a sidecar file beside
    the ``.npz`` (``cache=True``) and only read back afterwards -- what a DataLoader worker
    does instead of shipping the 5 MB dense tables (``FSADataset.__getitem__``,
    dataset_reader.py:30-40).  A stale sidecar (older than the ``.npz``, or packed for another
    ABI version) is rebuilt."""
    import os

    from .lattice import LatticeBatch

    side = packed_sidecar(npz_fname, which)
    if cache and os.path.exists(side) and os.path.getmtime(side) >= os.path.getmtime(npz_fname):
        try:
            return LatticeBatch.load(side)
        except ValueError:
            pass
    with np.load(npz_fname, allow_pickle=False) as l:
        em, tr = l[f"{which}_emission"], l[f"{which}_transition"]
    lat = LatticeBatch.from_dense(em[None], tr[None])
    if cache:
        tmp = side + f".{os.getpid()}.tmp.npz"
        lat.save(tmp)
        os.replace(tmp, side)  # workers may race for the same example
    return lat


def collate_packed(examples, device=None):
    """Batch of packed examples -> one LatticeBatch (``collate``, dataset_reader.py:175-186,
    without the pad-id padding rows: every lattice keeps its own row count)."""
    from .lattice import LatticeBatch

    return LatticeBatch.concat(list(examples), device=device)


class DevicePrefetcher:
    """Yields the packed batches of an iterable on the device, with the host-to-device copy of the
    next batch running on a side stream while the current one is being used -- the role of the
    DataLoader's ``pin_memory`` + ``.to(device, non_blocking=True)`` in the reference's trainer
    (a packed BASELINE batch is ~240 MB: ~5 ms over PCIe, a hundred sweep steps' worth)."""

    def __init__(self, batches, device):
        import torch

        self.batches = batches
        self.device = torch.device(device)

    def __iter__(self):
        import torch

        side = torch.cuda.Stream(self.device)

        def stage(b):
            pinned = b.pin_memory()
            with torch.cuda.stream(side):
                d = pinned.to(self.device, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(side)
            return d, ev, pinned  # the pinned copy must outlive the transfer

        it = iter(self.batches)
        try:
            nxt = stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur = nxt
            try:
                nxt = stage(next(it))
            except StopIteration:
                nxt = None
            torch.cuda.current_stream(self.device).wait_event(cur[1])
            for t in cur[0]._t.values():
                if t is not None:
                    t.record_stream(torch.cuda.current_stream(self.device))
            yield cur[0]
