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
    return (npz_fname[: -len(".npz")] if npz_fname.endswith(".npz") else npz_fname) + f".{which}.nfstpk"


def load_packed(npz_fname: str, which: str = "num", cache: bool = True, verify: bool = True, validate: bool = True):
    """The ``num`` (or ``denom``) lattice of one example as a packed one-lattice batch.  The
    packer runs the first time an example is seen; its output is kept in a sidecar file beside
    the ``.npz`` (``cache=True``) and only read back afterwards -- what a DataLoader worker
    does instead of shipping the 5 MB dense tables (``FSADataset.__getitem__``,
    dataset_reader.py:30-40).  A stale sidecar (older than the ``.npz``, packed for another ABI
    version, or failing its checksums / ``nfst_validate_batch``) is rebuilt."""
    import os

    from .lattice import LatticeBatch

    side = packed_sidecar(npz_fname, which)
    if cache and os.path.exists(side) and os.path.getmtime(side) >= os.path.getmtime(npz_fname):
        try:
            return LatticeBatch.load(side, verify=verify, validate=validate)
        except ValueError:
            pass
    with np.load(npz_fname, allow_pickle=False) as l:
        em, tr = l[f"{which}_emission"], l[f"{which}_transition"]
    lat = LatticeBatch.from_dense(em[None], tr[None])
    if cache:
        tmp = side + f".{os.getpid()}.tmp"
        lat.save(tmp)
        os.replace(tmp, side)  # workers may race for the same example
    return lat


class PackedReader:
    """Reads packed sidecars into a ring of ``slots`` reusable byte buffers (no fresh pages to fault in for every file:
    2.7 ms per 0.9 MB file through a memory map, 0.6 ms through a fresh buffer, ~0.15 ms into a warm one on the boxes
    of this pool).  A loaded batch is valid until its slot comes round again: gather ``slots`` examples at most --
    typically one step's batch, ``collate_packed(..., arena=...)`` -- before reading on."""

    def __init__(self, slots: int, verify: bool = True, validate: bool = True):
        self.slots, self.verify, self.validate = max(1, int(slots)), verify, validate
        self.bufs, self.turn = [None] * self.slots, 0

    def load(self, fname: str):
        import os

        from .lattice import LatticeBatch

        i = self.turn % self.slots
        self.turn += 1
        need = os.path.getsize(fname)
        if self.bufs[i] is None or self.bufs[i].size < need:
            self.bufs[i] = np.empty(int(need * 1.25) + 4096, dtype=np.uint8)
        return LatticeBatch.load(fname, verify=self.verify, validate=self.validate, buffer=self.bufs[i])


def collate_packed(examples, device=None, arena=None):
    """Batch of packed examples -> one LatticeBatch (``collate``, dataset_reader.py:175-186,
    without the pad-id padding rows: every lattice keeps its own row count).  ``arena``: a
    ``lattice.HostArena`` whose page-locked buffers receive the batch (one memcpy per array and part)."""
    from .lattice import LatticeBatch

    return LatticeBatch.concat(list(examples), device=device, arena=arena)


class DevicePrefetcher:
    """Yields the packed batches of an iterable on the device, with the host-to-device copy of the
    next batch running on a side stream while the current one is being used -- the role of the
    DataLoader's ``pin_memory`` + ``.to(device, non_blocking=True)`` in the reference's trainer
    (a packed BASELINE batch is ~240 MB: ~5 ms over PCIe, a hundred sweep steps' worth).
    The items of ``batches`` are packed batches or lists of packed parts (one-example sidecars: they are
    concatenated straight into the staging buffers).  Staging goes through ``depth`` page-locked arenas that are
    pinned once and reused (round 2 pinned every batch on the fly: 0.47 GB/s instead of the PCIe rate)."""

    def __init__(self, batches, device, depth: int = 3):
        import torch

        self.batches = batches
        self.device = torch.device(device)
        self.depth = max(2, int(depth))
        self._arenas = None  # page-locked once, on first use, and kept for every later pass over `batches`

    def __iter__(self):
        import torch

        from .lattice import HostArena, LatticeBatch

        side = torch.cuda.Stream(self.device)
        if self._arenas is None:
            self._arenas = [HostArena(pin=True) for _ in range(self.depth)]
        arenas = self._arenas
        busy = [None] * self.depth  # the event after which an arena's buffers may be overwritten
        turn = [0]

        def stage(item):
            i = turn[0] % self.depth
            turn[0] += 1
            if busy[i] is not None:
                busy[i].synchronize()
            parts = list(item) if isinstance(item, (list, tuple)) else [item]
            pinned = LatticeBatch.concat(parts, arena=arenas[i])  # (one part: a plain copy into the arena)
            with torch.cuda.stream(side):
                d = pinned.to(self.device, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(side)
            busy[i] = ev
            return d, ev

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
            main = torch.cuda.current_stream(self.device)
            ck = cur[0].chunks  # (a batch of deep, narrow lattices: its chunked programs and their scratch were allocated on the side stream too)
            for t in list(cur[0]._t.values()) + ([] if ck is None else list(ck._t.values()) + [ck.ws]):
                if t is not None:
                    t.record_stream(main)
            yield cur[0]
