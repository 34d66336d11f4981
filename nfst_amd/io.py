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
