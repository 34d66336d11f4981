#!/usr/bin/env python3
"""bench.py -- lattice-arcs/sec of the forward-backward (log-Z) hot path.

    python bench.py --gpus N --steps K --warmup W

A step is one pass of ``nfst_forward_backward`` (alpha sweep, beta sweep, exact
log Z, arc posteriors; float32 arithmetic in an extended-exponent probability
semiring) over one batch of synthetic lattices that is already resident in HBM:
BASELINE.json configs[1], 256 lattices of ~2k states / ~20k arcs per GPU
(``nfst_amd.synth.bench_batch``, seeds 1234+i).  With N > 1 every rank owns its
own 256 lattices (weak scaling, no data-path collective) and the only exchange
is the RCCL all-reduce of the scalar loss sum(log Z) at the end of each step.

Rank 0 prints one JSON line: value = lattice arcs processed by all ranks per
second; ``roofline`` = algorithmic HBM bytes (32 B/arc +
24 B/state) per launch / average kernel duration from HIP events, against the
8 TB/s HBM3E peak; ``cpu_baseline`` = the CPU oracle's float64 forward-backward
(oracle/nfst_oracle.c, OpenMP over lattices) on the same batch on this box's
host cores, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def pmc_traffic_bytes():
    """HBM bytes per k_forward_backward launch from the committed rocprofv3 PMC passes
    (profiles/collect.sh: FETCH_SIZE and WRITE_SIZE in separate runs of this benchmark, KiB per
    dispatch).  gfx950 correction from MI355X_MICROARCH.md section HBM: FETCH_SIZE counts 64 B per
    128-B request of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        return (2.0 * d["FETCH_SIZE"]["mean_per_dispatch_KiB"] + d["WRITE_SIZE"]["mean_per_dispatch_KiB"]) * 1024.0, \
            os.path.basename(files[-1])
    except Exception:
        return None, None


def cpu_baseline(lats, theta, budget_s=12.0):
    """The oracle (a port of the reference's path-sum semantics, float64) timed on
    the host cores over the same lattices; repeated until ~budget_s of work."""
    from oracle import oracle as O
    from nfst_amd import synth

    cores = os.cpu_count() or 1
    n_rows, arc_off, src, label, dst, w = synth.batch_arcs(lats)
    dp = int((src != dst).sum())
    O.forward_backward_batch(n_rows, arc_off, src, label, dst, w, theta, n_threads=cores)  # warm-up
    reps, t0 = 0, time.perf_counter()
    while True:
        O.forward_backward_batch(n_rows, arc_off, src, label, dst, w, theta, n_threads=cores)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or reps >= 200:
            break
    out = {"value": dp * reps / dt, "unit": "lattice-arcs/s", "cores": cores, "kind": "port",
           "sample": f"{len(lats)} lattices ({dp} arcs) x {reps} passes, float64 log-semiring forward-backward, "
                     f"OpenMP over lattices, {dt:.1f} s"}
    # the reference's own algorithm (compute_beta_parallel: dense [S,S] frontier loop, H=8) on one
    # small lattice -- its cost grows ~S^3, a 2k-state lattice takes minutes
    try:
        small = synth.layered_lattice(1234, n_states=300, avg_degree=8.0, vocab=64, width=8, span=4)
        _, tr = small.dense()
        rng = np.random.default_rng(0)
        H = 8
        emb, Wx = rng.normal(size=(64, H)).astype(np.float32), (0.3 * rng.normal(size=(H, H))).astype(np.float32)
        Wh, W, bias = np.zeros((H, H), np.float32), rng.normal(size=H).astype(np.float32), np.zeros(H, np.float32)
        t0 = time.perf_counter()
        O.beta_dense_frontier(tr, emb, Wx, Wh, W, bias)
        dt = time.perf_counter() - t0
        out["reference_dense_frontier"] = {"value": int((small.src != small.dst).sum()) / dt, "unit": "lattice-arcs/s",
                                           "cores": 1, "sample": f"one lattice S=300, beta sweep only, {dt:.2f} s"}
    except Exception as e:  # the extra line is informational
        out["reference_dense_frontier"] = {"error": str(e)}
    return out


def measured_copy_gbs(dev, mib: int = 1024, reps: int = 10) -> float:
    """What a plain device-to-device copy of 1 GiB reaches on this card (read + write bytes per
    second), the figure SURVEY.md section 8d wants beside the 8 TB/s nameplate peak."""
    src = torch.empty(mib << 20, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    dst.copy_(src)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        dst.copy_(src)
    b.record()
    torch.cuda.synchronize()
    return 2.0 * src.numel() * reps / (a.elapsed_time(b) * 1e-3) / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--lattices-per-gpu", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--slots", type=int, default=0, help="arc slots per lane of a tile (0 = packer's choice)")
    ap.add_argument("--no-compact", action="store_true", help="32-bit arc records instead of the compact tile format")
    ap.add_argument("--group-mode", type=int, default=0, help="0 = packer's choice, 1 = narrow groups, 2 = wide groups")
    ap.add_argument("--width", type=int, default=16, help="layer width of the synthetic lattices")
    ap.add_argument("--graph", action="store_true", help="replay a HIP graph of the step instead of launching from Python "
                    "(measured slower on ROCm 7.2: 67.8 vs 59.2 us per step)")
    ap.add_argument("--event-every", type=int, default=8, help="HIP events around runs of n back-to-back launches")
    ap.add_argument("--torch-sum", action="store_true", help="reduce the loss with torch.sum instead of the kernel's fused total")
    ap.add_argument("--mode", default="fb", choices=["fb", "fb_sweeps_only", "bwd"],
                    help="fb = the benchmark; the others are diagnostics (not the BASELINE metric)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs an MI355X (no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        # leave 56 KiB of every CU's LDS free (costs 0.8 % of the step: shallower rings) so that
        # RCCL's all-reduce kernel can run beside a sweep workgroup instead of taking a whole CU
        # away from the next launch (a sweep workgroup otherwise owns its CU's LDS)
        os.environ.setdefault("NFST_LDS_RESERVE_KB", "56")
        import torch.distributed as dist
        # RCCL ("nccl" on ROCm) over xGMI; NFST_BENCH_BACKEND=gloo only to rehearse the multi-rank
        # code path on a box where several ranks must share one GPU
        backend = os.environ.get("NFST_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from nfst_amd import ops, synth
    from nfst_amd.lattice import LatticeBatch
    from nfst_amd.distributed import all_reduce_loss

    B = args.lattices_per_gpu
    lats = synth.bench_batch(B, first_seed=1234 + rank * B, width=args.width)
    theta_np = synth.label_scores(1, 256)
    t0 = time.perf_counter()
    lat = LatticeBatch.from_synth(lats, slots_per_lane=args.slots, group_mode=args.group_mode, no_compact=args.no_compact)
    pack_s = time.perf_counter() - t0
    lat = lat.to(dev)
    theta = torch.from_numpy(theta_np).to(dev)
    arcs = int(lat.n_dp_arcs.sum())
    alg_bytes = lat.algorithmic_bytes("forward_backward")

    state = {"out": None, "i": 0}
    # sum_b log Z[b] comes out of the kernel itself (atomic adds into one of three rotating slots):
    # the loss of a step needs no reduction kernel.  TRIPLES slot triples take turns, so a step's
    # slot is cleared only 2 * TRIPLES steps later: its all-reduce over ranks has that long to
    # finish and no launch ever queues behind a collective.
    TRIPLES = 4
    total = torch.zeros(3 * TRIPLES, dtype=torch.float64, device=dev)
    fused = args.mode != "bwd" and not args.torch_sum and not args.graph  # a captured launch has one fixed slot

    def slot_of(i):
        return i % TRIPLES, (i // TRIPLES) % 3

    def run():
        # outputs are allocated by the first call and overwritten afterwards (steady state)
        j, slot = slot_of(state["i"])
        state["i"] += 1
        kw = dict(total=total[3 * j:3 * j + 3], total_slot=slot) if fused else {}
        if args.mode == "fb":
            state["out"] = ops.forward_backward(lat, theta, want_alpha_beta=True, want_posterior=True, out=state["out"], **kw)
        elif args.mode == "fb_sweeps_only":
            state["out"] = ops.forward_backward(lat, theta, want_alpha_beta=False, want_posterior=False, out=state["out"], **kw)
        else:
            return ops.backward(lat, theta, want_logbeta=False)
        return state["out"]

    import collections
    pending = collections.deque()

    def reduce_loss(r):
        """sum of log Z of this rank, all-reduced over ranks (RCCL) asynchronously and in place, in
        the slot the kernel left it in: the collectives of the last steps overlap the sweeps of the
        next ones; the oldest is waited for (long finished) before its slot comes up for clearing"""
        j, slot = slot_of(state["i"] - 1)
        loss = total[3 * j + slot:3 * j + slot + 1] if fused else r.logz64.sum()
        if world > 1:
            while len(pending) >= (2 * TRIPLES - 2 if fused else 1):
                pending.popleft().wait()
            loss, work = all_reduce_loss(loss, async_op=True, inplace=fused)
            pending.append(work)
        return loss

    # --graph: a step is launched by replaying a HIP graph of the forward-backward kernel (the
    # engine allocates nothing and keeps no state, so one warm-up call makes it capturable); the
    # reduction of the loss and its all-reduce follow on the same stream.  Default: the kernel is
    # launched from Python every step (faster here).
    use_graph = args.graph
    graph = None
    if use_graph:
        run()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                run()
        torch.cuda.current_stream().wait_stream(side)

    def launch():
        if graph is not None:
            graph.replay()
            return state["out"] if args.mode != "bwd" else bwd_out["r"]
        return run()

    bwd_out = {"r": None}
    if args.mode == "bwd" and use_graph:
        raise SystemExit("--mode bwd is a diagnostic: not with --graph")

    def step():
        r = launch()
        return r, reduce_loss(r)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    # kernel duration: HIP events on the launch stream around every run of `span` back-to-back
    # launches of the timed region, divided by span (nothing else runs on that stream, so this is
    # an upper bound of the average kernel duration; an event pair around every single launch adds
    # ~2.5 us of command gaps to each and reads 5 % high against rocprofv3's kernel trace)
    span = max(1, min(args.event_every, args.steps))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps // span)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        g, k = divmod(i, span)
        if k == 0 and g < len(ev):
            ev[g][0].record()
        r = launch()
        if k == span - 1 and g < len(ev):
            ev[g][1].record()
        loss = reduce_loss(r)
    while pending:
        pending.popleft().wait()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) / span
    total_arcs = arcs
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        a = torch.tensor([arcs], dtype=torch.float64, device=dev)
        dist.all_reduce(a, op=dist.ReduceOp.SUM)
        total_arcs = int(a.item())
    if rank == 0:
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        tbytes, tsrc = pmc_traffic_bytes() if (args.mode == "fb" and B == 256 and args.width == 16) else (None, None)
        out = {
            "metric": "lattice-arcs/sec forward-backward (log-Z)",
            "value": total_arcs * args.steps / dt,
            "unit": "lattice-arcs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: {B} synthetic lattices per GPU, ~2k states / ~20k arcs "
                                   f"(layer width {args.width}), alpha+beta+logZ+arc posteriors",
                       "lattices_per_gpu": B, "arcs_per_gpu": arcs, "vocab": 256,
                       "max_depth": int(lat.depth.max()), "max_tiles": int(lat.max_tiles), "loss": -float(loss.item()), "host_pack_s": pack_s,
                       "launch": "hip_graph_replay" if use_graph else "python",
                       "loss_reduction": "fused in the kernel (atomic adds)" if fused else "torch.sum",
                       "lds_reserve_kb": int(os.environ.get("NFST_LDS_RESERVE_KB", "0") or 0)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None if tbytes is None else tbytes / (kern_ms * 1e-3) / 1e9,
                         "traffic_bytes_per_launch": tbytes, "traffic_source": tsrc,
                         "kernel": "k_forward_backward", "kernel_ms": kern_ms, "algorithmic_bytes": alg_bytes},
        }
        out["roofline"]["hbm_copy_measured"] = measured_copy_gbs(dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(lats, theta_np, args.cpu_budget)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
