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
        """samplers.py:162-180: the columns up to the first one that is pad everywhere are walked;
        marks equal to 0 are dropped, the rest is left-aligned and followed by pad.  Like the
        reference's loop, a row whose walked part ends in a dropped 0 keeps that 0 right after its
        last mark (it differs from pad only when pad != 0).  Vectorised: a stable sort replaces the loop."""
        assert len(sequences.shape) == 2
        pad = self.model.__pad__
        all_pad = (sequences == pad).all(dim=0)
        idx = torch.nonzero(all_pad)
        stop = int(idx[0]) if idx.numel() > 0 else sequences.shape[1] - 1
        seq = sequences[:, : stop + 1]
        keep = seq != 0
        order = torch.sort((~keep).to(torch.int8), dim=1, stable=True).indices
        packed = torch.gather(seq, 1, order)
        n_keep = keep.sum(dim=1, keepdim=True)
        cols = torch.arange(stop + 1, device=sequences.device)[None, :]
        packed = torch.where(cols < n_keep, packed, torch.full_like(packed, pad))
        if pad != 0:  # the last write of the reference's loop at position n_keep is that trailing 0
            packed = torch.where((cols == n_keep) & (seq[:, -1:] == 0), torch.zeros_like(packed), packed)
        return packed.contiguous()


class ProposalSampler(Sampler):
    """``Sampler.stateful_sample`` for a *learned* proposal (samplers.py:182-335).  The network
    stays outside: ``score_fn(hx, inp) -> (new_hx, logits [N, V])`` is the reference's recurrent
    cell + output layer (``left_to_right_score`` without its masks).  Everything on the lattice
    side of a step -- emission and bos/pad/eos legality masks, the optional next-state value
    gather (``use_beta``, scorers.py:581-593), the insertion / length penalties (654-677), temperature,
    the categorical draw or the forced symbol, its log probability and the state advance -- is one
    ``nfst_proposal_step`` launch.  ``log_q`` is differentiable in the network's logits and in ``values``
    (the reference's ``Categorical.log_prob``, samplers.py:256-273; tune_proposal, lightning.py:339-406).

    ``penalties``: ``dict(insertion_mark=..., insert_threshold=2, insert_penalty=1000.0, length_threshold=0,
    length_penalty=1000.0)`` switches on the penalties of ``FSAGRUScorer`` (constructor defaults,
    scorers.py:930-933); ``beta_from_previous_state=True`` (the reference's order) gathers ``values`` from
    the transition row of the state before the previous symbol was consumed."""

    def __init__(self, model: LatticeScorer, score_fn, penalties: Optional[Dict] = None,
                 beta_from_previous_state: bool = True, sync_every: int = 8):
        super().__init__(model)
        self.score_fn = score_fn
        self.penalties = penalties
        self.beta_from_previous_state = beta_from_previous_state
        self.sync_every = max(1, int(sync_every))  # steps between two looks at the device-side "every walker has ended" counters

    def stateful_sample(self, batch_size: int, to_evaluate: Optional[torch.Tensor] = None, hx=None,
                        temperature: float = 1.0, values: Optional[torch.Tensor] = None,
                        uniforms: Optional[torch.Tensor] = None, return_zs: bool = False):
        """``(log_q [N], samples [N, T], hx)``, or ``(log_q, hx)`` when ``to_evaluate`` is given
        (``(log_q, zs, hx)`` with ``return_zs``), as samplers.py:322-335.  ``uniforms`` [T, N]
        replace the generator (tests); ``values`` is row-indexed like ``compute_log_beta``."""
        m = self.model
        lat = m._lat()
        dev = lat.device
        assert batch_size == lat.n_lattices * m.k  # samplers.py:146-150
        evaluate_only = to_evaluate is not None
        if evaluate_only:
            assert int(to_evaluate.shape[0]) == batch_size
            pad_col = torch.full((batch_size, 1), m.__pad__, dtype=torch.int64, device=dev)
            padded = torch.cat([to_evaluate.to(device=dev, dtype=torch.int64), pad_col], dim=1)  # samplers.py:208-218
        inp = torch.full((batch_size,), m.__bos__, dtype=torch.int64, device=dev)
        # the implicit bos is consumed first (scorers.py:230-231, metadata "state" after inp0 = bos)
        prev_state = torch.zeros(batch_size, dtype=torch.int64, device=dev)
        state = ops.step(lat, prev_state, inp, k=m.k)
        pen = None
        if self.penalties is not None:
            pen = ops.StepPenalties(batch_size, lat.vocab, dev, **self.penalties)
        # Outputs of all steps live in buffers allocated once; "has every walker ended" is a device word per step that
        # the kernel counts into, read back every CHECK steps: one host-device synchronisation per CHECK steps instead
        # of one per step (the reference tests all_reached_eos at every step, samplers.py:288-290).  Steps that ran
        # past the first all-pad step are dropped again, hx included: results are those of the step-by-step loop.
        CHECK = self.sync_every
        T = (m.max_length + 1) if not evaluate_only else min(m.max_length + 1, padded.shape[1])
        sym_all = torch.empty((T, batch_size), dtype=torch.int64, device=dev)
        nxt_all = torch.empty((T, batch_size), dtype=torch.int64, device=dev)
        logq_all = torch.empty((T, batch_size), dtype=torch.float32, device=dev)
        logz_all = torch.empty((T, batch_size), dtype=torch.float32, device=dev)
        not_pad = torch.zeros(T, dtype=torch.int32, device=dev)
        logqs, logzs, hxs = [], [], []
        n_steps, hard_cut = 0, True
        for timestep in range(T):
            hx, logits = self.score_fn(hx, inp)
            r = ops.proposal_step(lat, state, logits, k=m.k, inp=inp, values=values, pad=m.__pad__, bos=m.__bos__,
                                  eos=m.__eos__, has_to_end=(timestep + 1) > m.max_length, temperature=temperature,
                                  uniforms=None if (evaluate_only or uniforms is None) else uniforms[timestep],
                                  forced=padded[:, timestep] if evaluate_only else None,
                                  value_state=prev_state if (values is not None and self.beta_from_previous_state) else None,
                                  penalties=pen, length=timestep + 1,
                                  out=(sym_all[timestep], logq_all[timestep], logz_all[timestep], nxt_all[timestep]),
                                  not_pad=not_pad[timestep:timestep + 1])
            logqs.append(r.logq)
            logzs.append(r.logz)
            hxs.append(hx)
            prev_state, state, inp = state, r.next_state, r.symbol
            n_steps = timestep + 1
            if n_steps % CHECK == 0 or n_steps == T:
                first = n_steps - (CHECK if n_steps % CHECK == 0 else n_steps % CHECK)
                ended = (not_pad[first:n_steps] == 0).nonzero()
                if ended.numel():
                    n_steps = first + int(ended[0]) + 1  # the first step at which every symbol was pad
                    hard_cut = False
                    break
        hx = hxs[n_steps - 1] if n_steps else hx
        zero = torch.zeros(batch_size, dtype=torch.float32, device=dev)
        log_q = torch.stack(logqs[:n_steps]).sum(dim=0) if n_steps else zero
        zs = torch.stack(logzs[:n_steps]).sum(dim=0) if n_steps else zero
        prefixes = [sym_all[t] for t in range(n_steps)]
        if hard_cut and not evaluate_only:
            raise Exception("a sample did not end within max_length + 1 steps")  # the reference raises here too (samplers.py:299-302)
        if evaluate_only:
            return (log_q, zs, hx) if return_zs else (log_q, hx)
        prefixes.pop(-1)  # the trailing all-pad column (samplers.py:304-307)
        samples = torch.stack(prefixes, dim=1) if prefixes else torch.empty(batch_size, 0, dtype=torch.int64, device=dev)
        return log_q, samples, hx
