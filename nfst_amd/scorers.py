"""Host-side mirror of the reference's lattice scorer interface.

``LatticeScorer`` keeps the method names, argument meaning and error behaviour of
``FSAGRUScorer`` (/root/reference/src/modules/scorers.py:604-1054) for the lattice
part of that class -- ``set_masks``, ``set_k``, ``compute_beta``,
``update_fsa_state``, ``mask_out_invalid`` -- and of ``WFSTScorer`` (1663-1687) for
the arc weights ("an arc's weight is solely decided by the mark on it"): one
learnable score ``theta[mark]``.  Every method is a thin call into the HIP engine
(nfst_amd.ops); nothing is computed on the CPU.  The recurrent proposal network
of the reference (GRUCell, queries) is out of scope.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch

from . import ops
from .lattice import LatticeBatch


class LatticeScorer(torch.nn.Module):
    def __init__(self, vocab_size: int, pad: int = 0, bos: int = 1, eos: int = 2, max_length: int = 400,
                 theta: Optional[torch.Tensor] = None, k: int = 1):
        super().__init__()
        assert vocab_size > 3  # scorers.py:223
        self.vocab_size = vocab_size
        self.__pad__, self.__bos__, self.__eos__ = pad, bos, eos
        self.max_length = max_length
        self.k = k
        init = torch.zeros(vocab_size) if theta is None else torch.as_tensor(theta, dtype=torch.float32).clone()
        self.theta = torch.nn.Parameter(init)
        self.lattice: Optional[LatticeBatch] = None
        self.use_beta = True

    # ------------------------------------------------------------ tables
    def set_masks(self, emission: torch.Tensor, transition: torch.Tensor):
        """scorers.py:877-885.  The dense tables are packed once (arcs + tile
        programs); K samples share them, nothing is expanded."""
        assert len(emission.shape) == 3
        assert len(transition.shape) == 3
        with torch.no_grad():
            self.lattice = LatticeBatch.from_dense(emission, transition, device=self.theta.device)
        return self

    def set_lattice(self, lattice: LatticeBatch):
        self.lattice = lattice.to(self.theta.device)
        return self

    def set_k(self, k: int):
        """scorers.py:887-918 materialises K copies of the tables; here K is only a
        launch parameter."""
        self.k = int(k)

    def _lat(self) -> LatticeBatch:
        assert self.lattice is not None, "set_masks() first"
        return self.lattice

    # ------------------------------------------------------------ beta sweep
    def compute_log_beta(self) -> torch.Tensor:
        """log beta ``[B*k, S+1]`` (rows unreachable from state 0 are -inf)."""
        lat = self._lat()
        r = ops.backward(lat, self.theta.detach())
        return lat.rows_view(r.logbeta).repeat_interleave(self.k, dim=0)

    def compute_beta(self) -> torch.Tensor:
        """``[B*k, S+1]`` in the probability domain like scorers.py:858-875 returns
        (float32: overflows to inf beyond log beta ~ 88, as the reference does)."""
        return torch.exp(self.compute_log_beta())

    def log_z(self) -> torch.Tensor:
        """Differentiable exact log Z per lattice ``[B]``; d/d theta = expected mark counts."""
        return ops.log_z(self._lat(), self.theta)

    # ------------------------------------------------------------ per-step gathers
    def update_fsa_state(self, updated: torch.Tensor, prev_states: torch.Tensor) -> torch.Tensor:
        """scorers.py:683-690."""
        return ops.step(self._lat(), prev_states, updated, k=self.k)

    def mask_out_invalid(self, inp: torch.Tensor, metadata: Dict[str, Any]) -> torch.Tensor:
        """scorers.py:1037-1054 on top of 314-338 (bos/pad/eos legality)."""
        assert self.lattice is not None
        assert "length" in metadata
        return ops.emission_mask(self._lat(), metadata["state"], k=self.k, inp=inp, pad=self.__pad__, bos=self.__bos__,
                                 eos=self.__eos__, has_to_end=metadata["length"] > self.max_length)

    def beta_logits(self, beta: torch.Tensor, state: torch.Tensor) -> torch.Tensor:
        """scorers.py:584-590: ``gather(beta, 1, transition_k[arange, state])``; beta ``[B*k, S+1]``
        with identical rows per lattice (as compute_beta returns) or ``[B, S+1]``."""
        lat = self._lat()
        rows = beta[:: self.k] if beta.shape[0] == lat.n_lattices * self.k and self.k > 1 else beta
        return ops.beta_logits(lat, rows.reshape(-1), state, k=self.k)

    # ------------------------------------------------------------ WFSTScorer
    def wfst_score(self, t: torch.Tensor) -> torch.Tensor:
        """scorers.py:1685-1687: sum over non-pad marks of theta[mark]."""
        assert len(t.shape) == 2
        sc = self.theta[t]
        return torch.where(t == self.__pad__, torch.zeros_like(sc), sc).sum(dim=1)

    def forward(self, sequence: torch.Tensor, **kwargs):
        return self.wfst_score(sequence)


class NeuralBetaScorer(LatticeScorer):
    """The ``use_beta=True`` part of ``FSAGRUScorer`` (scorers.py:954-970): parameters ``Wh``, ``Wx``
    [H, H], ``W`` [1, H], ``beta_bias`` [H] and the mark ``embeddings`` [V, H], initialised as
    there; ``compute_beta`` is the message-passing sweep of scorers.py:692-751 / 753-856 on the HIP
    engine (``ops.backward_neural``) and follows the per-sample semantics -- parallel arcs between
    a state pair count once each."""

    def __init__(self, hid_dim: int, vocab_size: int, **kw):
        super().__init__(vocab_size, **kw)
        self.hid_dim = hid_dim
        self.embeddings = torch.nn.Embedding(vocab_size, hid_dim)
        self.Wh = torch.nn.Parameter(torch.empty(hid_dim, hid_dim))
        self.Wx = torch.nn.Parameter(torch.empty(hid_dim, hid_dim))
        self.W = torch.nn.Parameter(torch.empty(1, hid_dim))
        for p in (self.Wh, self.Wx, self.W):
            torch.nn.init.xavier_uniform_(p)
        self.beta_bias = torch.nn.Parameter(torch.zeros(hid_dim))

    def compute_beta_hat(self):
        """``(log beta [B*k, S+1], beta_hat [B*k, S+1, H])``.  Differentiable in the five parameters
        (``tune_proposal`` trains them through it): under autograd the call keeps the forward workspace for the
        backward pass -- 2 B max_rows (H + 1) floats, 1.1 GB at H = 256 on the BASELINE batch.  A caller that only
        samples wraps the call in ``torch.no_grad()`` and nothing is kept."""
        lat = self._lat()
        r = ops.backward_neural(lat, self.embeddings.weight, self.Wx, self.Wh, self.W, self.beta_bias)
        return (lat.rows_view(r.log_beta).repeat_interleave(self.k, dim=0),
                lat.rows_view(r.beta_hat).repeat_interleave(self.k, dim=0))

    def compute_log_beta(self) -> torch.Tensor:
        return self.compute_beta_hat()[0]
