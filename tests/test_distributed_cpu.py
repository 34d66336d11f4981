"""Multi-rank path on CPU: lattices are sharded by arc count,
each rank scores its own shard, the only exchange is the all-reduce of the scalar
loss (gloo here, RCCL on the GPUs).  The per-lattice scores come from the CPU oracle
-- this test is about the sharding and the collectives, not about the kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nfst_amd import synth
from nfst_amd.distributed import all_reduce_loss, gather_logz, shard_lattices


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_lat, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    lats = [synth.layered_lattice(100 + i, n_states=40 + 13 * (i % 5), avg_degree=4.0, vocab=32, width=4, span=3)
            for i in range(n_lat)]
    theta = synth.label_scores(3, 32)
    shards = shard_lattices([l.n_arcs for l in lats], world)
    mine = shards[rank]
    z = torch.tensor([O.forward_backward(lats[i].n_rows, lats[i].src, lats[i].dst,
                                         theta[lats[i].label].astype(np.float64))["logZ"] for i in mine],
                     dtype=torch.float64)
    loss = all_reduce_loss(-z.sum())
    full = gather_logz(z, mine, n_lat)
    if rank == 0:
        np.savez(out_path, loss=loss.numpy(), full=full.numpy(), shard_sizes=np.array([len(s) for s in shards]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_lattices_is_balanced_and_deterministic():
    n_arcs = [100, 900, 400, 400, 50, 700, 300, 1000, 20, 650]
    a = shard_lattices(n_arcs, 3)
    assert a == shard_lattices(n_arcs, 3)
    assert sorted(i for s in a for i in s) == list(range(len(n_arcs)))
    loads = [sum(n_arcs[i] for i in s) for s in a]
    assert max(loads) - min(loads) <= max(n_arcs)
    assert shard_lattices([5, 5], 4) == [[0], [1], [], []]


def test_two_rank_loss_allreduce(tmp_path):
    from oracle import oracle as O
    world, n_lat = 2, 11
    out = os.path.join(tmp_path, "r0.npz")
    mp.spawn(_worker, args=(world, _free_port(), n_lat, out), nprocs=world, join=True)
    d = np.load(out)
    lats = [synth.layered_lattice(100 + i, n_states=40 + 13 * (i % 5), avg_degree=4.0, vocab=32, width=4, span=3)
            for i in range(n_lat)]
    theta = synth.label_scores(3, 32)
    ref = np.array([O.forward_backward(l.n_rows, l.src, l.dst, theta[l.label].astype(np.float64))["logZ"] for l in lats])
    assert np.allclose(d["full"], ref, atol=1e-12)
    assert abs(float(d["loss"]) + ref.sum()) < 1e-9
    assert d["shard_sizes"].sum() == n_lat and d["shard_sizes"].min() >= 4
