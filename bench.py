#!/usr/bin/env python3
"""bench.py -- lattice-arcs/sec of the forward-backward (log-Z) hot path.

    python bench.py --gpus N --steps K --warmup W

A step is one pass of ``nfst_forward_backward`` (alpha sweep, beta sweep, exact
log Z, arc posteriors; float32 arithmetic in an extended-exponent probability
semiring) over one batch of synthetic lattices that is already resident in HBM
(``nfst_amd.synth.bench_batch``, lattices of ~2k states / ~20k arcs, seeds 1234+i).

Workload: N = 1 runs BASELINE.json configs[1] (256 lattices on the GPU); N > 1 runs
configs[3]'s shape, 1024 lattices per GPU (8192 over 8 GPUs).  Every rank owns its own
lattices (weak scaling, no data-path collective); the only exchange is the RCCL
all-reduce of the scalar loss sum(log Z) at the end of each step.  The other per-GPU
batch size is timed as well and reported under ``config.aux`` ("per_gpu_256" /
"per_gpu_1024") at every N, so that scaling can be read at equal per-GPU work.

``python bench.py --gpus N`` without WORLD_SIZE in the environment starts its own N rank
processes (fresh children, before this process touches the GPU); under
``torch.distributed.run`` it is one of the ranks.

The timed loop ROTATES over several distinct resident batches (four at 256 lattices per GPU): one
batch moves ~131 MB per launch, less than the 256 MiB Infinity Cache, so replaying a single batch
would measure the die-level cache, not HBM (MI355X_MICROARCH.md: a line stays resident only while
everything touched between two of its uses fits in ~256 MiB).  With the rotation ~390 MB of other
lattices and outputs pass between two uses of a line: every launch starts cold.  The replayed figure
is reported beside it (``roofline.kernel_ms_replay``); ``value``, ``roofline.frac`` and the rocprofv3
summaries of the same command are the cold ones.

Rank 0 prints one JSON line: value = lattice arcs processed by all ranks per
second; ``roofline`` = algorithmic HBM bytes (32 B/arc +
24 B/state) per launch / average kernel duration from HIP events, against the
8 TB/s HBM3E peak (``frac``), and beside it ``frac_hw`` = the HBM bytes the PMC counters saw per
launch (profiles/<tag>_pmc_traffic.json, taken on this same command) / the same duration / peak; ``cpu_baseline`` = the CPU oracle's float64 forward-backward
(oracle/nfst_oracle.c, OpenMP over lattices) on the same batch on this box's
host cores, rank 0 at N=1 only.  ``config.aux`` (N = 1): timed lines for the other
BASELINE configs -- [2] 64 SNIPS-shaped lattices, [4] sampling K=16 + Viterbi with
float32 / bfloat16-rounded scores --, for shallower / deeper lattices of the same size,
and for the step with caller-supplied per-arc scores.
"""
import argparse
import collections
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


# ----------------------------------------------------------------------------- self-launch
def self_launch(n: int) -> int:
    """Start n fresh rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set),
    wait for all of them; a failing rank ends the others.  Runs before anything in this process
    touches the GPU -- a process that has initialised the GPU is never re-executed.  Only the
    standard library is used here."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in live:  # the exact children started above, by PID
                    q.terminate()
    return rc


# ----------------------------------------------------------------------------- helpers
TRAFFIC_FILES = ("r03_pmc_traffic.json", "r02_pmc_traffic.json")  # newest first; each names its workload


def pmc_traffic_bytes(lattices_per_gpu: int, rotate: int):
    """HBM bytes per k_forward_backward launch from the committed rocprofv3 PMC passes
    (profiles/collect.sh: FETCH_SIZE and WRITE_SIZE in separate runs of this benchmark, KiB per
    dispatch).  gfx950 correction from MI355X_MICROARCH.md section HBM: FETCH_SIZE counts 64 B per
    128-B request of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact.
    The file is chosen by name, newest round first, and must have been taken on this workload (the
    number of lattices per GPU it records; files of round 2 carry none and were taken at 256, replaying
    one batch).  Returns (bytes, file name, "rotating" | "replay")."""
    for name in TRAFFIC_FILES:
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            continue
        if int(d.get("lattices_per_gpu", 256)) != lattices_per_gpu:
            continue
        mode = "rotating" if int(d.get("rotate", 1)) > 1 else "replay"
        if (rotate > 1) != (mode == "rotating"):
            continue
        return (2.0 * d["FETCH_SIZE"]["mean_per_dispatch_KiB"] + d["WRITE_SIZE"]["mean_per_dispatch_KiB"]) * 1024.0, name, mode
    return None, None, None


def host_cores() -> int:
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (a GPU box
    can show 256 logical CPUs to a job that owns a 16-CPU share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(lats, theta, budget_s=12.0):
    """The oracle (a port of the reference's path-sum semantics, float64) timed on
    the host cores over the same lattices; repeated until ~budget_s of work."""
    import numpy as np
    from oracle import oracle as O
    from nfst_amd import synth

    cores = host_cores()
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
    out = {"value": dp * reps / dt, "unit": "lattice-arcs/s", "cores": cores, "host_logical_cpus": os.cpu_count(), "kind": "port",
           "sample": f"{len(lats)} lattices ({dp} arcs) x {reps} passes, float64 log-semiring forward-backward, "
                     f"OpenMP over lattices, {dt:.1f} s"}
    # cpu_ref_dense: the reference's own algorithm
    # (compute_beta_parallel: dense [S,S] cell table, frontier loop, H = 8 label-only weights; cost
    # ~S^3) restated in C, on the first two lattices of the batch at their full size (S ~ 2000) with
    # parallel arcs removed (the reference's algorithm never finishes a lattice that has them,
    # SURVEY 8a-3); the rows of the cell table are shared by OpenMP threads like torch's intra-op
    # threads share the reference's einsums.  The reference's Python takes ~200 s per such lattice
    # on 8 cores.
    try:
        rng = np.random.default_rng(0)
        H = 8
        V = lats[0].vocab
        emb, Wx = rng.normal(size=(V, H)).astype(np.float32), (0.3 * rng.normal(size=(H, H))).astype(np.float32)
        Wh, W, bias = np.zeros((H, H), np.float32), np.full(H, -2.3 / H, np.float32), np.full(H, 4.0, np.float32)
        arcs_done, t0 = 0, time.perf_counter()
        sizes = []
        for l in lats[:2]:
            l2 = synth.without_parallel_arcs(l)
            _, tr = l2.dense()
            O.beta_dense_frontier(tr, emb, Wx, Wh, W, bias, n_threads=cores)
            arcs_done += int((l2.src != l2.dst).sum())
            sizes.append(l2.n_rows - 1)
        dt = time.perf_counter() - t0
        out["cpu_ref_dense"] = {"value": arcs_done / dt, "unit": "lattice-arcs/s", "cores": cores,
                                "sample": f"{len(sizes)} lattices, S = {sizes}, beta sweep only, dense-frontier algorithm of "
                                          f"scorers.py:753-856 restated in C (float32, H = 8), {dt:.2f} s"}
    except Exception as e:  # the extra line is informational
        out["cpu_ref_dense"] = {"error": str(e)}
    return out


def measured_copy_gbs(dev, mib: int = 1024, reps: int = 10) -> float:
    """What a plain device-to-device copy of 1 GiB reaches on this card (read + write bytes per
    second), the figure SURVEY.md section 8d wants beside the 8 TB/s nameplate peak."""
    import torch
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


def time_op(fn, iters: int, warmup: int = 5) -> float:
    """ms per call of ``fn`` (launches on torch's current stream, which is the stream the ops use)."""
    import torch
    for _ in range(warmup):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


class Stepper:
    """The benchmark step over one resident batch: launch + loss reduction (+ RCCL all-reduce).

    sum_b log Z[b] comes out of the kernel itself (atomic adds into one of three rotating slots):
    the loss of a step needs no reduction kernel.  TRIPLES slot triples take turns, so a step's
    slot is cleared only 2 * TRIPLES steps later: its all-reduce over ranks has that long to
    finish and no launch ever queues behind a collective."""
    TRIPLES = 4

    def __init__(self, lat, theta, dev, world, mode="fb", fused=True, arc_scores=None):
        import torch
        self.torch = torch
        self.lat, self.theta, self.world, self.mode, self.fused = lat, theta, world, mode, fused
        self.arc_scores = arc_scores
        self.total = torch.zeros(3 * self.TRIPLES, dtype=torch.float64, device=dev)
        self.out, self.i, self.plans = None, 0, None
        self.pending = collections.deque()

    def slot_of(self, i):
        return i % self.TRIPLES, (i // self.TRIPLES) % 3

    def launch(self):
        from nfst_amd import ops
        j, slot = self.slot_of(self.i)
        self.i += 1
        if self.mode == "bwd":
            return ops.backward(self.lat, self.theta, arc_scores=self.arc_scores, want_logbeta=False)
        # the steady state of a training loop: same batch, same score tensors, outputs overwritten -- one prepared launch per
        # slot triple (ops.ForwardBackwardLaunch: a ctypes call per step; the plain wrapper's ~40 us of host work per call is
        # longer than the kernel and left the GPU idle between launches)
        if self.plans is None:
            full = self.mode == "fb"
            self.plans = []
            for t in range(self.TRIPLES if self.fused else 1):  # (every plan writes the same output tensors)
                self.plans.append(ops.ForwardBackwardLaunch(self.lat, self.theta, arc_scores=self.arc_scores, want_alpha_beta=full,
                                                            want_posterior=full, out=self.plans[0].out if self.plans else None,
                                                            total=self.total[3 * t:3 * t + 3] if self.fused else None))
        self.out = self.plans[j if self.fused else 0](slot if self.fused else 0)
        return self.out

    def reduce_loss(self, r):
        """sum of log Z of this rank, all-reduced over ranks (RCCL) asynchronously and in place, in
        the slot the kernel left it in: the collectives of the last steps overlap the sweeps of the
        next ones; the oldest is waited for (long finished) before its slot comes up for clearing"""
        from nfst_amd.distributed import all_reduce_loss
        j, slot = self.slot_of(self.i - 1)
        loss = self.total[3 * j + slot:3 * j + slot + 1] if self.fused else r.logz64.sum()
        if self.world > 1:
            while len(self.pending) >= (2 * self.TRIPLES - 2 if self.fused else 1):
                self.pending.popleft().wait()
            loss, work = all_reduce_loss(loss, async_op=True, inplace=self.fused)
            self.pending.append(work)
        return loss

    def drain(self):
        while self.pending:
            self.pending.popleft().wait()


class Rotor:
    """Round-robin over steppers of distinct resident batches (see the module docstring: cold launches)."""

    def __init__(self, steppers):
        self.steppers, self.i = list(steppers), 0
        self.cur = self.steppers[0]

    def launch(self):
        self.cur = self.steppers[self.i % len(self.steppers)]
        self.i += 1
        return self.cur.launch()

    def reduce_loss(self, r):
        return self.cur.reduce_loss(r)

    def drain(self):
        for s in self.steppers:
            s.drain()

    def arcs_of_steps(self, first: int, n: int) -> int:
        """lattice arcs of steps first .. first + n - 1 of the rotation"""
        return sum(int(self.steppers[(first + k) % len(self.steppers)].lat.n_dp_arcs.sum()) for k in range(n))


def timed_region(stepper, steps, warmup, world, dev, event_every=8, launch=None):
    """W untimed warm-up steps, then exactly `steps` steps bracketed by barrier + synchronize on
    both sides.  Returns (wall seconds -- max over ranks --, list of per-launch kernel ms, one per
    event window, last loss).  Kernel duration: HIP events on the launch stream around every run of
    `span` back-to-back launches, divided by span (nothing else runs on that stream, so this is an
    upper bound of the average kernel duration; an event pair around every single launch adds
    ~2.5 us of command gaps to each and reads 5 % high against rocprofv3's kernel trace)."""
    import torch
    import torch.distributed as dist
    launch = launch or stepper.launch
    loss = None
    for _ in range(warmup):
        loss = stepper.reduce_loss(launch())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    span = max(1, min(event_every if steps >= 32 else 4, steps))  # (a 20-step run still gives five windows)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps // span)]
    t0 = time.perf_counter()
    for i in range(steps):
        g, k = divmod(i, span)
        if k == 0 and g < len(ev):
            ev[g][0].record()
        r = launch()
        if k == span - 1 and g < len(ev):
            ev[g][1].record()
        loss = stepper.reduce_loss(r)
    stepper.drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    windows = [a.elapsed_time(b) / span for a, b in ev]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, windows, loss


def sum_over_ranks(x, world, dev):
    if world == 1:
        return x
    import torch
    import torch.distributed as dist
    a = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(a, op=dist.ReduceOp.SUM)
    return int(a.item())


def window_stats(windows):
    import numpy as np
    w = np.asarray(windows, dtype=np.float64)
    return {"n_windows": int(w.size), "min_ms": float(w.min()), "median_ms": float(np.median(w)), "max_ms": float(w.max())}


def global_shard(G, world, rank, width):
    """The job's G lattices (seeds 1234 + i) split over the ranks as the engine's callers would
    (SURVEY 8e): every rank draws the G lattice sizes -- the state count S_i comes from the lattice's seed
    alone, the arc count is ~10 S_i --, computes ``distributed.shard_lattices`` (greedy LPT, the same answer on
    every rank, nothing is exchanged) and generates only its own lattices.  Returns (indices, lattices)."""
    import numpy as np
    from nfst_amd import synth
    from nfst_amd.distributed import shard_lattices
    S = [int(np.random.default_rng((1234 + i) ^ 0x5EED).integers(1800, 2201)) for i in range(G)]
    mine = shard_lattices([10 * x for x in S], world)[rank]
    return mine, [synth.layered_lattice(1234 + i, n_states=S[i], width=width) for i in mine]


def build_batch(B, rank, width, dev, first_seed=None, **pack):
    from nfst_amd import synth
    from nfst_amd.lattice import LatticeBatch
    lats = synth.bench_batch(B, first_seed=(1234 + rank * B) if first_seed is None else first_seed, width=width)
    t0 = time.perf_counter()
    lat = LatticeBatch.from_synth(lats, **pack)
    pack_s = time.perf_counter() - t0
    return lats, lat.to(dev), pack_s


def aux_single_gpu(dev, theta256, steps):
    """Timed lines for what the headline does not cover (N = 1): the other BASELINE configs and
    the shape / score variants of configs[1].  Not the metric; reported under config.aux."""
    import numpy as np
    import torch
    from nfst_amd import ops, synth
    from nfst_amd.lattice import LatticeBatch

    aux = {}
    iters = max(10, min(steps, 100))

    def fb_line(lat, theta, arc_scores=None):
        out = {"o": None}

        def f():
            out["o"] = ops.forward_backward(lat, theta, arc_scores=arc_scores, out=out["o"])
        # (the better of two runs: the GPU sat idle while the host built the batch, and once in a while the first window of a
        # small batch ran at the clocks of an idle chip -- 107 us instead of 71 for the SNIPS-shaped batch on one box)
        ms = min(time_op(f, iters, warmup=10), time_op(f, iters))
        arcs = int(lat.n_dp_arcs.sum())
        return {"lattices": lat.n_lattices, "arcs": arcs, "max_depth": int(lat.depth.max()), "ms_per_step": ms,
                "arcs_per_s": arcs / (ms * 1e-3), "roofline_frac": lat.algorithmic_bytes() / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}

    # configs[1] variants: shallow / deep lattices of the same size; caller-supplied per-arc scores
    for width in (64, 4):
        _, lat, _ = build_batch(256, 0, width, dev)
        aux[f"configs1_width{width}"] = fb_line(lat, theta256)
    _, lat, _ = build_batch(256, 0, 16, dev)
    asc = torch.randn(lat.total_arcs, device=dev) * 0.1
    aux["configs1_with_arc_scores"] = fb_line(lat, theta256, arc_scores=asc)
    aux["configs1_beta_only"] = {"ms_per_step": time_op(lambda: ops.backward(lat, theta256, want_logbeta=False), iters)}
    del lat, asc
    # configs[2]: 64 SNIPS-shaped lattices (long and narrow: up to ~750 token positions)
    lats = synth.snips_shaped_batch(64)
    th = torch.from_numpy(synth.label_scores(64, lats[0].vocab, mean=-1.5, std=0.8)).to(dev)
    lat = LatticeBatch.from_synth(lats, device=dev)
    aux["configs2_snips_shaped_b64"] = fb_line(lat, th)
    aux["configs2_snips_shaped_b64"]["viterbi_ms"] = time_op(lambda: ops.viterbi(lat, th), iters)
    # (deep and narrow: the packer cut chunked programs on the way to the device; the general kernels on the same batch beside it)
    from nfst_amd import _lib
    aux["configs2_snips_shaped_b64"]["flavour"] = "chunked" if lat.chunks is not None else "general"
    if lat.chunks is not None:  # (lattices the chunked kernels handed back to the general ones: none expected on these scores)
        aux["configs2_snips_shaped_b64"]["handed_back_to_general_kernels"] = int(lat.chunks.flagged().sum())
    aux["configs2_snips_shaped_b64"]["beta_only_ms"] = time_op(lambda: ops.backward(lat, th, want_logbeta=False), iters)
    with _lib.tuning(chunked=0):
        aux["configs2_snips_shaped_b64"]["general_kernels_ms_per_step"] = fb_line(lat, th)["ms_per_step"]
        aux["configs2_snips_shaped_b64"]["general_kernels_beta_only_ms"] = time_op(lambda: ops.backward(lat, th, want_logbeta=False), iters)
    # batch size 1: the reference's decoder scores one lattice at a time (src/decode/decoder.py:77-79)
    one = LatticeBatch.from_synth(lats[:1], device=dev)
    aux["decode_b1_snips_shaped"] = dict(fb_line(one, th), viterbi_ms=time_op(lambda: ops.viterbi(one, th), iters))
    # ... and one deep, narrow machine (two states per position, 750 levels): the chunked flavour on two CUs
    deep1 = LatticeBatch.from_synth([synth.layered_lattice(4242, n_states=1500, avg_degree=3.0, vocab=lats[0].vocab, width=2, span=1, max_degree=40)], device=dev)
    aux["decode_b1_deep_narrow"] = dict(fb_line(deep1, th), flavour="chunked" if deep1.chunks is not None else "general")
    with _lib.tuning(chunked=0):
        aux["decode_b1_deep_narrow"]["general_kernels_ms_per_step"] = fb_line(deep1, th)["ms_per_step"]
    del deep1
    lats1 = synth.bench_batch(1, first_seed=1234)
    one = LatticeBatch.from_synth(lats1, device=dev)
    aux["decode_b1_configs1_lattice"] = dict(fb_line(one, theta256), viterbi_ms=time_op(lambda: ops.viterbi(one, theta256), iters))
    del one
    # configs[4]: posterior sampling K = 16 + Viterbi on transliteration-shaped (edit) lattices,
    # float32 scores vs the same scores rounded to bfloat16 (float32 accumulation either way)
    rng = np.random.default_rng(4)
    V = 64
    lats = []
    for i in range(256):
        x = rng.integers(3, V, size=int(rng.integers(6, 15))).tolist()
        y = rng.integers(3, V, size=int(rng.integers(6, 15))).tolist()
        lats.append(synth.edit_lattice(x, y, vocab=V, seed=100 + i))
    lat = LatticeBatch.from_synth(lats, device=dev)
    th32 = torch.from_numpy(synth.label_scores(12, V, mean=-1.0, std=0.7)).to(dev)
    thbf = th32.to(torch.bfloat16).to(torch.float32)
    T = int(lat.depth.max()) + 1
    line = {"lattices": 256, "K": 16, "arcs": int(lat.n_dp_arcs.sum()), "max_len": T}
    for tag, th in (("f32", th32), ("bf16_scores", thbf)):
        line[f"sample_k16_ms_{tag}"] = time_op(lambda: ops.sample_paths(lat, th, 16, max_len=T, seed=1), iters)
        line[f"viterbi_ms_{tag}"] = time_op(lambda: ops.viterbi(lat, th), iters)
    z32 = ops.backward(lat, th32, want_logbeta=False).logz64
    zbf = ops.backward(lat, thbf, want_logbeta=False).logz64
    line["max_abs_logz_diff_bf16_scores"] = float((z32 - zbf).abs().max())
    line["walks_per_s_f32"] = 256 * 16 / (line["sample_k16_ms_f32"] * 1e-3)
    aux["configs4_sampling_viterbi"] = line
    return aux


# ----------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--lattices-per-gpu", type=int, default=0,
                    help="0 = BASELINE: 256 at N = 1 (configs[1]), 1024 at N > 1 (configs[3]: 8192 over 8 GPUs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux", action="store_true", help="only the headline line (no config.aux measurements)")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--slots", type=int, default=0, help="arc slots per lane of a tile (0 = packer's choice)")
    ap.add_argument("--no-compact", action="store_true", help="32-bit arc records instead of the compact tile format")
    ap.add_argument("--group-mode", type=int, default=0, help="0 = packer's choice, 1 = narrow groups, 2 = wide groups")
    ap.add_argument("--width", type=int, default=16, help="layer width of the synthetic lattices")
    ap.add_argument("--graph", action="store_true", help="replay a HIP graph of the step instead of launching from Python "
                    "(measured slower on ROCm 7.2: 67.8 vs 59.2 us per step)")
    ap.add_argument("--event-every", type=int, default=8, help="HIP events around runs of n back-to-back launches (4 when steps < 32)")
    ap.add_argument("--no-replay", action="store_true", help="skip the extra loop that replays one resident batch (profiles/collect.sh: "
                    "the rocprofv3 summary then averages cold launches only)")
    ap.add_argument("--rotate", type=int, default=0,
                    help="distinct resident batches the timed loop rotates over (0 = as many as it takes to push more than "
                         "the 256 MiB Infinity Cache between two uses of a line: 4 at 256 lattices per GPU, 1 from 1024 on)")
    ap.add_argument("--torch-sum", action="store_true", help="reduce the loss with torch.sum instead of the kernel's fused total")
    ap.add_argument("--mode", default="fb", choices=["fb", "fb_sweeps_only", "bwd"],
                    help="fb = the benchmark; the others are diagnostics (not the BASELINE metric)")
    ap.add_argument("--arc-scores", action="store_true", help="tuning: caller-supplied per-arc scores (the autograd path) in every mode")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="no GPU, no kernel, no number: ranks rendezvous over gloo, pack a few lattices, all-reduce a "
                         "scalar and rank 0 prints a line with value null (tests of the launcher on a CPU box)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))  # before anything touches the GPU

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)

    import numpy as np
    import torch

    if args.rehearse_cpu:
        return rehearse_cpu(args, rank, world)
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

    from nfst_amd import synth

    B = args.lattices_per_gpu or (256 if world == 1 else 1024)
    pack = dict(slots_per_lane=args.slots, group_mode=args.group_mode, no_compact=args.no_compact)
    shard = None
    if world > 1:
        # the real sharding path: LPT over the job's B * world lattices, every rank packs its own shard
        from nfst_amd.lattice import LatticeBatch
        shard, lats = global_shard(B * world, world, rank, args.width)
        t0 = time.perf_counter()
        lat = LatticeBatch.from_synth(lats, **pack).to(dev)
        pack_s = time.perf_counter() - t0
    else:
        lats, lat, pack_s = build_batch(B, rank, args.width, dev, **pack)
    theta_np = synth.label_scores(1, 256)
    theta = torch.from_numpy(theta_np).to(dev)
    arcs = int(lat.n_dp_arcs.sum())
    alg_bytes = lat.algorithmic_bytes("forward_backward")

    fused = args.mode != "bwd" and not args.torch_sum and not args.graph  # a captured launch has one fixed slot
    asc = torch.randn(lat.total_arcs, device=dev) * 0.1 if args.arc_scores else None
    st = Stepper(lat, theta, dev, world, mode=args.mode, fused=fused, arc_scores=asc)
    # cold launches: rotate over distinct resident batches (module docstring).  One launch touches ~0.51 MB per
    # lattice (tile programs, canonical arcs, outputs); between two uses of a line the other batches of the rotation
    # must move more than the Infinity Cache holds.
    touched = 0.51e6 * B
    n_rot = args.rotate if args.rotate > 0 else (1 if (touched > 300e6 or args.graph) else int(min(8, 1 + -(-300e6 // touched))))
    steppers = [st]
    for r in range(1, n_rot):
        _, lat_r, _ = build_batch(B, rank, args.width, dev, first_seed=1234 + (world * r + rank) * B + 100000 * r, **pack)
        asc_r = torch.randn(lat_r.total_arcs, device=dev) * 0.1 if args.arc_scores else None
        steppers.append(Stepper(lat_r, theta, dev, world, mode=args.mode, fused=fused, arc_scores=asc_r))
    rotor = Rotor(steppers)

    # --graph: a step is launched by replaying a HIP graph of the forward-backward kernel (the
    # engine allocates nothing, so one warm-up call makes it capturable); the reduction of the loss
    # and its all-reduce follow on the same stream.  Default: the kernel is launched from Python
    # every step (faster here).
    launch = None
    if args.graph:
        if args.mode == "bwd":
            raise SystemExit("--mode bwd is a diagnostic: not with --graph")
        st.launch()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                st.launch()
        torch.cuda.current_stream().wait_stream(side)

        def launch():
            graph.replay()
            return st.out

    dt, windows, loss = timed_region(rotor, args.steps, args.warmup, world, dev, args.event_every, launch)
    arcs_timed = rotor.arcs_of_steps(args.warmup, args.steps) if launch is None else arcs * args.steps
    total_arcs_timed = sum_over_ranks(arcs_timed, world, dev)
    kern_ms = float(np.mean(windows))
    alg_bytes = float(np.mean([s_.lat.algorithmic_bytes("forward_backward") for s_ in steppers]))
    # the same step replaying ONE resident batch (what rounds 1-2 reported): its ~131 MB per launch fit the Infinity Cache
    replay_ms = None
    if n_rot > 1 and not args.no_replay:
        _, win_r, _ = timed_region(st, max(8, min(args.steps, 64)), 4, world, dev, args.event_every)
        replay_ms = float(np.mean(win_r))
    replay_frac = None if replay_ms is None else lat.algorithmic_bytes() / (replay_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    # multi-rank: the per-lattice scores reassembled in job order on every rank (decode-style callers), checked against
    # the oracle on a sample of this rank's own lattices; stragglers made visible (per-rank kernel time, arcs per rank)
    sharding = None
    if world > 1:
        import torch.distributed as dist
        from nfst_amd.distributed import gather_logz
        from oracle import oracle as O
        G = B * world
        full = gather_logz(st.out.logz64, shard, G)
        err = 0.0
        for j in range(0, len(lats), max(1, len(lats) // 4)):
            l = lats[j]
            o = O.forward_backward(l.n_rows, l.src, l.dst, theta_np[l.label].astype(np.float64))
            err = max(err, abs(float(full[shard[j]]) - o["logZ"]))
        stats = torch.tensor([kern_ms, float(arcs), float(lat.n_lattices), err], dtype=torch.float64, device=dev)
        allst = [torch.zeros_like(stats) for _ in range(world)]
        dist.all_gather(allst, stats)
        allst = torch.stack(allst).cpu().numpy()
        sharding = {"global_batch": G, "policy": "greedy LPT by estimated arcs (10 x states), computed on every rank",
                    "lattices_per_rank": [int(x) for x in allst[:, 2]], "arcs_per_rank": [int(x) for x in allst[:, 1]],
                    "lpt_imbalance_max_over_mean_arcs": float(allst[:, 1].max() / allst[:, 1].mean()),
                    "kernel_ms_per_rank_min": float(allst[:, 0].min()), "kernel_ms_per_rank_max": float(allst[:, 0].max()),
                    "gathered_logz_max_abs_err_vs_oracle_sample": float(allst[:, 3].max()),
                    "gathered_logz_finite": bool(torch.isfinite(full).all())}
        assert sharding["gathered_logz_max_abs_err_vs_oracle_sample"] <= 1e-5, sharding
    del rotor, steppers[1:]

    # the other per-GPU batch size of the BASELINE configs, every rank, same protocol, fewer steps
    aux = {}
    default_shape = args.lattices_per_gpu == 0 and args.mode == "fb" and args.width == 16 and not args.graph and not args.arc_scores
    if default_shape and not args.no_aux:
        B2 = 1024 if B == 256 else 256
        key = {256: "per_gpu_256", 1024: "per_gpu_1024"}
        del st
        lats2, lat2, _ = build_batch(B2, rank, args.width, dev, **pack)
        st2 = Stepper(lat2, theta, dev, world, fused=True)
        steps2 = max(8, min(args.steps, 100))
        dt2, win2, _ = timed_region(st2, steps2, min(args.warmup, 10), world, dev, args.event_every)
        arcs2 = sum_over_ranks(int(lat2.n_dp_arcs.sum()), world, dev)
        k2 = float(np.mean(win2))
        aux[key[B2]] = {"workload": f"{B2} lattices per GPU ({'configs[3]: 8192 over 8 GPUs' if B2 == 1024 else 'configs[1]'})",
                        "value": arcs2 * steps2 / dt2, "unit": "lattice-arcs/s", "n_gpus": world, "steps": steps2,
                        "ms_per_step": dt2 / steps2 * 1e3, "kernel_ms": k2,
                        "roofline_frac": lat2.algorithmic_bytes() / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBS}
        aux[key[B]] = "the headline line"
        del st2, lat2, lats2
        if world == 1:
            try:
                aux.update(aux_single_gpu(dev, theta, args.steps))
            except Exception as e:  # informational lines must not cost the headline
                aux["error"] = f"{type(e).__name__}: {e}"

    if rank == 0:
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        tbytes, tsrc, tmode = pmc_traffic_bytes(B, n_rot) if (args.mode == "fb" and args.width == 16 and not args.arc_scores) else (None, None, None)
        shape = "BASELINE configs[1]" if B == 256 else ("BASELINE configs[3] (8192 lattices over 8 GPUs)" if B == 1024 else "custom")
        out = {
            "metric": "lattice-arcs/sec forward-backward (log-Z)",
            "value": total_arcs_timed / dt,
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
            "config": {"workload": f"{shape}: {B} synthetic lattices per GPU, ~2k states / ~20k arcs "
                                   f"(layer width {args.width}), alpha+beta+logZ+arc posteriors",
                       "lattices_per_gpu": B, "lattices_total": B * world, "arcs_per_gpu": arcs, "vocab": 256,
                       "resident_batches_rotated": n_rot, "sharding": sharding,
                       "cache_state": ("cold: %d distinct resident batches take turns, ~%.0f MB touched between two uses of a line "
                                       "(Infinity Cache: 256 MiB)" % (n_rot, (n_rot - 1) * touched / 1e6)) if n_rot > 1 else
                                      ("one resident batch, ~%.0f MB touched per launch" % (touched / 1e6)),
                       "max_depth": int(lat.depth.max()), "max_tiles": int(lat.max_tiles), "loss": -float(loss.item()),
                       "host_pack_s": pack_s,
                       "launch": "hip_graph_replay" if args.graph else "prepared launch (ops.ForwardBackwardLaunch: one ctypes call per step)",
                       "loss_reduction": "fused in the kernel (atomic adds)" if fused else "torch.sum",
                       "lds_reserve_kb": int(os.environ.get("NFST_LDS_RESERVE_KB", "0") or 0),
                       "aux": aux},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "frac_is": "algorithmic bytes (SURVEY 8d: 32 B/arc + 24 B/state) / kernel_ms_cold / peak",
                         # bytes the memory system really moved per launch (PMC counters), on the same duration
                         "traffic": None if tbytes is None else tbytes / (kern_ms * 1e-3) / 1e9,
                         "frac_hw": None if tbytes is None else tbytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "traffic_bytes_per_launch": tbytes, "traffic_source": tsrc, "traffic_taken_on": tmode,
                         "kernel": "k_forward_backward", "kernel_ms": kern_ms,
                         "kernel_ms_cold": kern_ms if n_rot > 1 else None, "kernel_ms_replay": replay_ms if n_rot > 1 else kern_ms,
                         "frac_replay": replay_frac,
                         "kernel_ms_windows": window_stats(windows),
                         "algorithmic_bytes": alg_bytes},
        }
        out["roofline"]["hbm_copy_measured"] = measured_copy_gbs(dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(lats[:256], theta_np, args.cpu_budget)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def rehearse_cpu(args, rank, world):
    """The launcher, the rank plumbing and the sharding path on a box without a GPU: gloo rendezvous, LPT sharding of
    a small job computed on every rank, the host packer on each rank's shard, the scalar all-reduce and ``gather_logz``
    (with a stand-in score per lattice: no kernel runs here).  Measures nothing and says so."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from nfst_amd import synth
    from nfst_amd.distributed import all_reduce_loss, gather_logz, shard_lattices
    from nfst_amd.lattice import LatticeBatch

    if world > 1:
        dist.init_process_group("gloo")
    G = 6 * world + 1
    sizes = [30 + 17 * (i % 5) + 9 * (i % 3) for i in range(G)]

    def make(i):
        return synth.layered_lattice(1234 + i, n_states=sizes[i], avg_degree=4.0, vocab=32, width=4, span=2)

    shards = shard_lattices([4 * x for x in sizes], world)  # every rank, same answer, nothing exchanged
    mine = shards[rank]
    lat = LatticeBatch.from_synth([make(i) for i in mine])
    score = torch.from_numpy(lat.n_dp_arcs.astype(np.float64))  # stand-in for log Z: the packed arc count of every lattice
    total = all_reduce_loss(score.sum().reshape(1))
    full = gather_logz(score, mine, G)
    arcs = torch.zeros(world, dtype=torch.float64)
    arcs[rank] = float(score.sum())
    if world > 1:
        dist.all_reduce(arcs)
        dist.barrier()
    if rank == 0:
        single = LatticeBatch.from_synth([make(i) for i in range(G)]).n_dp_arcs.astype(np.float64)  # one process, whole job
        print(json.dumps({"metric": "lattice-arcs/sec forward-backward (log-Z)", "value": None, "unit": "lattice-arcs/s",
                          "n_gpus": world, "rehearsal": "cpu: no kernel was run, nothing was measured",
                          "arcs_packed_all_ranks": int(total.item()),
                          "sharding": {"global_batch": G, "lattices_per_rank": [len(x) for x in shards],
                                       "lpt_imbalance_max_over_mean_arcs": float(arcs.max() / arcs.mean()),
                                       "gathered_matches_single_process": bool(np.array_equal(full.numpy(), single)),
                                       "sum_matches_single_process": bool(abs(float(total.item()) - single.sum()) < 1e-9)}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
