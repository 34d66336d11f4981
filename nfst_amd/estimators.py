"""Host-side mirror of ``Estimators`` (/root/reference/src/modules/estimatros.py:9-44)."""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from . import ops
from .samplers import Sampler


class Estimators:
    @staticmethod
    def _sample_and_score(batch_size, k_prime, proposal: Sampler, query_args, unnormalized_model):
        proposal.set_k(k_prime)
        log_q, samples = proposal.sample(batch_size=batch_size * k_prime, to_evaluate=None, query_args=query_args)
        log_q = log_q.reshape((batch_size, k_prime))
        unnormalized_log_p = unnormalized_model(proposal.stripping_pad(samples))
        samples = samples.reshape((batch_size, k_prime, -1))
        return log_q, unnormalized_log_p.reshape((batch_size, k_prime)), samples

    @staticmethod
    def batched_importance_sampling(batch_size, k_prime, proposal: Sampler, query_args, unnormalized_model, step):
        """estimatros.py:11-30: ``(log_q [B,K], log_w [B,K], samples [B,K,T])`` with
        ``log_w = log p~ - log_q.detach()`` (28)."""
        log_q, log_p, samples = Estimators._sample_and_score(batch_size, k_prime, proposal, query_args, unnormalized_model)
        return log_q, log_p - log_q.detach(), samples

    @staticmethod
    def iwae(proposal: Sampler, unnormalized_model: torch.nn.Module, batch_size: int, k: int, step,
             query_args: Optional[Dict] = None):
        """estimatros.py:33-44: ``(log_marginal [B], log_q [B,K], samples [B,K,T], log_w [B,K])``.
        With the exact posterior proposal and a per-mark model every log_w equals
        log p~(path) - (path score - log Z), i.e. the estimate has zero variance."""
        log_q, log_p, samples = Estimators._sample_and_score(batch_size, k, proposal, query_args, unnormalized_model)
        if log_p.requires_grad:
            # the reference's autograd semantics: gradient through log p~, log q detached
            log_w = log_p - log_q.detach()
            return torch.logsumexp(log_w, dim=1) - math.log(k), log_q, samples, log_w
        lm, log_w = ops.iwae(log_p, log_q.detach())
        return lm, log_q, samples, log_w

    @staticmethod
    def exact(proposal: Sampler):
        """The exact log-marginal the IWAE estimate converges to: log Z per lattice
        (differentiable through the model's per-mark scores)."""
        return proposal.model.log_z()
