"""Multi-GPU: lattices shard embarrassingly.

One process per GPU; each rank packs and sweeps its own lattices; the only
exchange is one all-reduce of the scalar loss sum(log Z) per step -- RCCL over
xGMI on the GPUs (``backend="nccl"`` is RCCL on ROCm), gloo in the CPU tests.
The reference's only multi-device mechanism is Lightning's implicit DDP
(/root/reference/src/trainer/tr_trainer.py:80-86); there is no explicit
collective to translate.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist


def shard_lattices(n_arcs: Sequence[int], world_size: int) -> List[List[int]]:
    """Greedy longest-processing-time partition of lattice indices by arc count:
    heaviest lattice first, always to the currently lightest rank.  Deterministic
    (ties broken by index / rank), every rank computes the same answer locally."""
    order = sorted(range(len(n_arcs)), key=lambda i: (-int(n_arcs[i]), i))
    loads = [0] * world_size
    shards: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda q: (loads[q], q))
        shards[r].append(i)
        loads[r] += int(n_arcs[i])
    return [sorted(s) for s in shards]


def local_log_z_sum(lat, theta, device=None) -> torch.Tensor:
    """sum_b log Z[b] of this rank's shard as a float64 scalar tensor.  ``shard_lattices`` hands a
    rank an empty shard when there are fewer lattices than ranks: such a rank passes ``lat=None``
    and contributes 0 to the all-reduce instead of launching on an empty batch (which the engine
    refuses with NFST_ERR_ARG) while its peers wait in the collective."""
    if lat is None or lat.n_lattices == 0:
        return torch.zeros((), dtype=torch.float64, device=device if device is not None else "cpu")
    from . import ops
    return ops.forward_backward(lat, theta, want_alpha_beta=False, want_posterior=False).logz64.sum()


def all_reduce_loss(loss: torch.Tensor, async_op: bool = False, inplace: bool = False):
    """Sum a scalar (float64 recommended) over ranks; 8 bytes, latency only.

    ``async_op=True`` returns ``(tensor, work)``: the collective runs on the backend's
    own stream and the caller's stream only waits at ``work.wait()`` -- the loss of
    step t is needed for logging, not by step t+1, so the next step's kernels need not
    queue behind the (latency-bound) all-reduce.  ``inplace=True`` reduces into ``loss`` itself
    (e.g. a slot of the kernel's fused total) instead of a copy."""
    work = None
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if not inplace:
            loss = loss.clone()
        work = dist.all_reduce(loss, op=dist.ReduceOp.SUM, async_op=async_op)
    return (loss, work) if async_op else loss


def gather_logz(local_logz: torch.Tensor, shard: Sequence[int], n_total: int) -> torch.Tensor:
    """Reassemble per-lattice log Z in the original batch order on every rank
    (used by decode-style callers that need each lattice's score, not the sum)."""
    out = torch.zeros(n_total, dtype=local_logz.dtype, device=local_logz.device)
    out[torch.as_tensor(list(shard), dtype=torch.long, device=local_logz.device)] = local_logz
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(out, op=dist.ReduceOp.SUM)
    return out
