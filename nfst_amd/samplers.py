"""Host-side mirror of ``Sampler`` (/root/reference/src/modules/samplers.py:21-335).

Same entry points (``set_masks``, ``set_k``, ``sample``, ``stripping_pad``,
``all_reached_eos``) and the same return contract -- ``(log_q [N], samples [N, T])``
with pad-terminated samples that start after the implicit ``bos`` and have the
trailing all-pad column removed (samplers.py:304-307).  The proposal is the exact
posterior over lattice paths under the model's per-mark scores (zero-variance
importance weights) instead of a learned GRU; the T-step Python loop of
``stateful_sample`` (243-297) is one kernel launch.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops
from .scorers import LatticeScorer


class Sampler:
    def __init__(self, model: LatticeScorer):
        self.model = model
        self.seed = 0

    @property
    def self_normalized(self):
        return True

    def set_masks(self, transition, emission):
        assert isinstance(self.model, LatticeScorer)
        self.model.set_masks(emission=emission, transition=transition)

    def set_k(self, k: int):
        self.model.set_k(k)

    def all_reached_eos(self, candidates: torch.Tensor):
        return (candidates == self.model.__pad__).all()

    def sample(self, batch_size: int, to_evaluate: Optional[torch.Tensor] = None, no_zs: bool = False,
               temperature: float = 1.0, query_args: Optional[Dict] = None, uniforms: Optional[torch.Tensor] = None):
        m = self.model
        lat = m._lat()
        assert batch_size == lat.n_lattices * m.k  # samplers.py:146-150
        assert temperature == 1.0, "the exact posterior proposal has no temperature"
        theta = m.theta.detach()
        if to_evaluate is not None:
            # forced scoring (samplers.py:208-218): log q of the given marks under the posterior
            assert int(to_evaluate.shape[0]) == batch_size
            marks = to_evaluate.reshape(lat.n_lattices, m.k, -1).to(torch.int32)
            bos = torch.full((lat.n_lattices, m.k, 1), m.__bos__, dtype=torch.int32, device=marks.device)
            tot, end = ops.score_paths(lat, theta, torch.cat([bos, marks], dim=2))
            z = ops.backward(lat, theta, want_logbeta=False).logz
            sink = torch.as_tensor(lat.sink, device=end.device, dtype=end.dtype)[:, None]
            log_q = torch.where(end == sink, tot - z[:, None], torch.full_like(tot, float("-inf")))
            return (log_q.reshape(-1),)
        max_len = min(int(lat.depth.max()) + 1, m.max_length + 2)
        self.seed += 1
        s = ops.sample_paths(lat, theta, m.k, max_len=max_len, uniforms=uniforms, seed=self.seed, pad=m.__pad__)
        # drop the implicit bos (samplers.py:230-231: inp0 = bos) and everything after the longest sample
        width = max(int(s.lengths.max()) - 1, 1)
        samples = s.paths[:, :, 1:1 + width].reshape(batch_size, width).to(torch.int64)
        return s.logq.reshape(-1), samples

    def stripping_pad(self, sequences: torch.Tensor) -> torch.Tensor:
        """samplers.py:162-180: drop marks equal to 0, left-align, cut after the first
        column that is pad everywhere.  (Vectorised: a stable sort replaces the loop.)"""
        assert len(sequences.shape) == 2
        pad = self.model.__pad__
        keep = sequences != 0
        order = torch.sort((~keep).to(torch.int8), dim=1, stable=True).indices
        packed = torch.gather(sequences, 1, order)
        n_keep = keep.sum(dim=1, keepdim=True)
        cols = torch.arange(sequences.shape[1], device=sequences.device)[None, :]
        packed = torch.where(cols < n_keep, packed, torch.full_like(packed, pad))
        all_pad = (sequences == pad).all(dim=0)
        idx = torch.nonzero(all_pad)
        stop = int(idx[0]) if idx.numel() > 0 else sequences.shape[1] - 1
        return packed[:, : stop + 1].contiguous()
