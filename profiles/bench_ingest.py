"""End-to-end ingest: example on disk -> log Z on the GPU (not the BASELINE metric).

Routes for N examples of the BASELINE shape (~2k states / ~20k arcs, V = 256), one host process:
  dense          the reference's .npz records (preprocess/tr.py:182-190): np.load (zlib) -> collate (pad-id padding)
                 -> host packer on the dense tables -> H2D -> forward-backward
  dense_device   the same records, collated tables moved to the GPU first (what Lightning does before set_masks,
                 lightning.py:417) -> the packer on the device
  sidecar        packed sidecars (written once per example): one-read load (checksums + nfst_validate_batch, or
                 trusted) -> nfst_concat_packed straight into page-locked staging -> H2D behind the previous batch
  arcs_device    12-byte-per-arc lists -> H2D -> the packer on the device
and the device packer alone on the BASELINE batch (256 lattices), the staging copy rate, the resident step for scale."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nfst_amd import io, ops, synth
from nfst_amd.lattice import HostArena, LatticeBatch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
BATCH = 32
dev = torch.device("cuda")
theta = torch.from_numpy(synth.label_scores(1, 256)).to(dev)
out = {"examples": N, "batch": BATCH}


def sync_time(fn, reps=1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, r


with tempfile.TemporaryDirectory() as d:
    lats = synth.bench_batch(N)
    files = []
    for i, l in enumerate(lats):
        em, tr = l.dense()
        f = os.path.join(d, f"ex{i}.npz")
        io.save_fsa_npz(f, (em, tr), (em[:2], tr[:2]), [1, 2], [1, 2])
        files.append(f)
    out["dense_npz_bytes_per_example"] = os.path.getsize(files[0])
    # ---- dense route (host packer) and dense_device route (tables to the GPU, packer there)
    t = {"load": 0.0, "collate": 0.0, "pack": 0.0, "h2d+fb": 0.0}
    td = {"h2d_tables": 0.0, "device_pack": 0.0, "fb": 0.0}
    # (warm-up: the first launch of a kernel pays for its code object and LDS opt-in)
    _w = io.collate([io.load_fsa_from_npz(f) for f in files[:2]], pad=synth.PAD)
    LatticeBatch.from_dense_device(torch.from_numpy(_w[0]).to(dev), torch.from_numpy(_w[1]).to(dev)); torch.cuda.synchronize()
    for b0 in range(0, N, BATCH):
        t0 = time.perf_counter(); recs = [io.load_fsa_from_npz(f) for f in files[b0:b0 + BATCH]]; t["load"] += time.perf_counter() - t0
        t0 = time.perf_counter(); col = io.collate(recs, pad=synth.PAD); t["collate"] += time.perf_counter() - t0
        t0 = time.perf_counter(); lat = LatticeBatch.from_dense(col[0], col[1]); t["pack"] += time.perf_counter() - t0
        dt, _ = sync_time(lambda: ops.forward_backward(lat.to(dev), theta)); t["h2d+fb"] += dt
        dt, (em_d, tr_d) = sync_time(lambda: (torch.from_numpy(col[0]).to(dev), torch.from_numpy(col[1]).to(dev))); td["h2d_tables"] += dt
        dt, lat_d = sync_time(lambda: LatticeBatch.from_dense_device(em_d, tr_d)); td["device_pack"] += dt
        dt, _ = sync_time(lambda: ops.forward_backward(lat_d, theta)); td["fb"] += dt
    out["dense_route_ms_per_example"] = {k: v / N * 1e3 for k, v in t.items()}
    out["dense_device_route_ms_per_example"] = {k: v / N * 1e3 for k, v in td.items()}
    out["dense_device_route_note"] = "load + collate as in the dense route; the tables are 5 MB per example: the copy is PCIe time"
    # ---- sidecar route: first pass writes the sidecars, then the steady state
    t0 = time.perf_counter(); [io.load_packed(f) for f in files]; out["sidecar_first_pass_ms_per_example"] = (time.perf_counter() - t0) / N * 1e3
    out["sidecar_bytes_per_example"] = os.path.getsize(io.packed_sidecar(files[0]))
    for tag, kw in (("checked", dict(verify=True, validate=True)), ("trusted", dict(verify=False, validate=False))):
        reader = io.PackedReader(N, **kw)  # (N slots: this script keeps every example's batch until the pipeline has run)
        side = [io.packed_sidecar(f) for f in files]
        [reader.load(f) for f in side]     # (warms the buffers)
        reader.turn = 0
        t0 = time.perf_counter(); parts = [reader.load(f) for f in side]; load_ms = (time.perf_counter() - t0) / N * 1e3
        groups = [parts[b0:b0 + BATCH] for b0 in range(0, N, BATCH)]
        arena = HostArena(pin=True)
        LatticeBatch.concat(groups[0], arena=arena)  # (pins the arena once)
        t0 = time.perf_counter()
        for g in groups:
            LatticeBatch.concat(g, arena=arena)
        concat_ms = (time.perf_counter() - t0) / N * 1e3
        pf = io.DevicePrefetcher(groups, dev)
        [None for _ in pf]  # (first pass: the staging arenas are allocated and page-locked)
        dt, z = sync_time(lambda: [ops.forward_backward(b, theta).logz64.sum() for b in pf])
        out[f"sidecar_route_{tag}_ms_per_example"] = {"load": load_ms, "concat_into_pinned": concat_ms,
                                                     "prefetch_concat_h2d_fb_pipeline": dt / N * 1e3}
        out[f"sidecar_route_{tag}_examples_per_s_one_host_process"] = 1.0 / ((load_ms + dt / N * 1e3) * 1e-3)
    # the staging copy alone: a packed batch from the page-locked arena to the device
    pinned = LatticeBatch.concat(groups[0], arena=arena)
    nbytes = sum(v.numel() * v.element_size() for v in pinned._t.values() if v is not None)
    dt, _ = sync_time(lambda: pinned.to(dev, non_blocking=True), reps=5)
    out["h2d_from_pinned_arena_GBps"] = nbytes / dt / 1e9
    out["packed_bytes_per_example"] = nbytes / len(groups[0])
    # ---- arc lists -> device packer
    t = {"h2d+device_pack": 0.0, "fb": 0.0}
    for b0 in range(0, N, BATCH):
        n_rows, arc_off, src, label, dst, w = synth.batch_arcs(lats[b0:b0 + BATCH])
        dt, lat_d = sync_time(lambda: LatticeBatch.from_arcs_device(n_rows, arc_off, src, label, dst, 256, device=dev)); t["h2d+device_pack"] += dt
        dt, _ = sync_time(lambda: ops.forward_backward(lat_d, theta)); t["fb"] += dt
    out["arcs_device_route_ms_per_example"] = {k: v / N * 1e3 for k, v in t.items()}
    out["arc_list_bytes_per_example"] = int(12 * lats[0].n_arcs)
# ---- the device packer on the BASELINE batch (arc lists already on the device), against the host packer
lats = synth.bench_batch(256)
n_rows, arc_off, src, label, dst, w = synth.batch_arcs(lats)
sd, ld, dd = (torch.from_numpy(x).to(dev) for x in (src, label, dst))
LatticeBatch.from_arcs_device(n_rows, arc_off, sd, ld, dd, 256, device=dev)  # warm-up (LDS opt-in, allocator)
dt, lat_d = sync_time(lambda: LatticeBatch.from_arcs_device(n_rows, arc_off, sd, ld, dd, 256, device=dev), reps=3)
out["device_pack_256_lattices_ms"] = dt * 1e3
t0 = time.perf_counter(); host = LatticeBatch.from_arcs(n_rows, arc_off, src, label, dst, 256); out["host_pack_256_lattices_ms"] = (time.perf_counter() - t0) * 1e3
for _ in range(5): ops.forward_backward(lat_d, theta)
dt, _ = sync_time(lambda: ops.forward_backward(lat_d, theta), reps=50)
out["forward_backward_ms_per_example_resident"] = dt / 256 * 1e3
print(json.dumps(out, indent=1))
