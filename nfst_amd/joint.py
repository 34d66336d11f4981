"""Host-side mirror of the trainer/decoder call sites of ``JointProb``
(/root/reference/src/modules/lightning.py): ``log_marginalize`` (408-440),
``forward`` (442-480) and ``decode_from_npz`` (647-658) keep their signatures and
return tuples; the body is the HIP lattice engine.  With ``exact=True`` (default)
``num_prob`` is the exact log-marginal log Z of the numerator lattice (what the
reference's IWAE estimate converges to) and ``best_sample`` is the Viterbi path;
with ``exact=False`` the IWAE path of the reference is followed with the exact
posterior as proposal.  The Lightning training loop, optimisers and the neural
scorers are out of scope.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import io, ops
from .estimators import Estimators
from .samplers import Sampler
from .scorers import LatticeScorer


class JointProb(torch.nn.Module):
    def __init__(self, vocab_size: int, pad: int = 0, bos: int = 1, eos: int = 2, k: int = 16, max_length: int = 400,
                 exact: bool = True, theta: Optional[torch.Tensor] = None):
        super().__init__()
        self.k = k
        self.exact = exact
        self.tilde_p = LatticeScorer(vocab_size, pad=pad, bos=bos, eos=eos, max_length=max_length, theta=theta)
        self.num_sampler = Sampler(self.tilde_p)
        self.step = 0

    def log_marginalize(self, emission, transition, proposal: Sampler, x=None, y=None, k: Optional[int] = None):
        """lightning.py:408-440: ``(log_marginalized [B], log_q [B,K], samples [B,K,T], log_w [B,K])``."""
        proposal.set_masks(transition=transition, emission=emission)
        out = Estimators.iwae(proposal, self.tilde_p, emission.shape[0], k, self.step, None)
        self.step += 1
        return out

    def forward(self, numerator_emission, numerator_transition, denom_emission, denom_transition, gs, ps,
                return_samples: bool = False):
        """lightning.py:442-480: ``(num_prob [B], denom_prob [B][, best_sample [T]])``; denom_prob is 0
        as in the reference (473)."""
        if self.exact:
            self.tilde_p.set_masks(emission=numerator_emission, transition=numerator_transition)
            num_prob = self.tilde_p.log_z()
            denom_prob = torch.zeros_like(num_prob)
            if not return_samples:
                return num_prob, denom_prob
            lat = self.tilde_p._lat()
            v = ops.viterbi(lat, self.tilde_p.theta.detach(), pad=self.tilde_p.__pad__)
            n = int(v.lengths.max())
            best = v.paths[:, 1:max(n, 2)].to(torch.int64)  # after the implicit bos, like sampler output
            return num_prob, denom_prob, (best[0] if best.shape[0] == 1 else best)
        num_prob, _, num_samples, log_w = self.log_marginalize(numerator_emission, numerator_transition,
                                                               self.num_sampler, x=gs, y=ps, k=self.k)
        denom_prob = torch.zeros_like(num_prob)
        if return_samples:
            log_w = log_w.squeeze()
            num_samples = num_samples.squeeze()
            best_sample = num_samples[torch.argmax(log_w)] if self.k > 1 else num_samples
            return num_prob, denom_prob, best_sample
        return num_prob, denom_prob

    def decode_from_npz(self, npz_path, vocab_size, pad):
        """lightning.py:647-658: ``(prob: float, mark: LongTensor[T])``."""
        single_batch = tuple(torch.from_numpy(_).unsqueeze(0) for _ in io.load_fsa_from_npz(npz_path))
        l = self.forward(*single_batch, return_samples=True)
        mark = l[2]
        prob = (l[0] - l[1]).flatten()[0].item()
        return prob, mark
