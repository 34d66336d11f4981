"""`python bench.py --gpus N` without WORLD_SIZE starts its own N ranks (fresh child processes,
before the parent touches the GPU).  Exercised here with two gloo ranks on the CPU in the
benchmark's rehearsal mode: rendezvous, LPT sharding, host packer, all-reduce, gather_logz -- no kernel, no number."""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=env, capture_output=True, text=True,
                          timeout=300)


def test_self_launch_two_gloo_ranks():
    p = _run("--gpus", "2", "--rehearse-cpu")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 prints the one JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] is None and "rehearsal" in d
    assert d["arcs_packed_all_ranks"] > 0
    # the sharding path of the multi-GPU benchmark: LPT shards computed on every rank, each rank packs its
    # own, the gathered per-lattice scores and the all-reduced sum equal the single-process result
    sh = d["sharding"]
    assert sh["global_batch"] == 13 and sum(sh["lattices_per_rank"]) == 13 and min(sh["lattices_per_rank"]) >= 5
    assert 1.0 <= sh["lpt_imbalance_max_over_mean_arcs"] < 1.15
    assert sh["gathered_matches_single_process"] and sh["sum_matches_single_process"]


def test_self_launch_three_ranks():
    p = _run("--gpus", "3", "--rehearse-cpu")
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 3 and d["sharding"]["gathered_matches_single_process"]


def test_failing_rank_fails_the_launch():
    if torch.cuda.is_available():
        return  # on a GPU box the ranks would run the real benchmark
    p = _run("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert p.returncode != 0  # no GPU here: every rank refuses ("no CPU fallback"), the launcher reports it
    assert "no CPU fallback" in p.stderr
