"""End-to-end ingest: example on disk -> log Z on the GPU (not the BASELINE metric).

Two routes for the reference's ``.npz`` records (preprocess/tr.py:182-190), N examples of the BASELINE shape:
  dense    np.load (zlib) -> collate (pad-id padding) -> host packer on the dense tables -> H2D -> forward-backward
           (what ``set_masks`` does when it is handed the reference's collated batch)
  sidecar  LatticeBatch.load of the packed sidecar (written once per example) -> concat -> pinned H2D behind the
           previous batch (io.DevicePrefetcher) -> forward-backward
Host stages are timed on ONE host process; the reference's trainer runs them in DataLoader workers."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nfst_amd import io, ops, synth
from nfst_amd.lattice import LatticeBatch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
BATCH = 32
dev = torch.device("cuda")
theta = torch.from_numpy(synth.label_scores(1, 256)).to(dev)
out = {"examples": N, "batch": BATCH}
with tempfile.TemporaryDirectory() as d:
    lats = synth.bench_batch(N)
    t0 = time.perf_counter()
    files = []
    for i, l in enumerate(lats):
        em, tr = l.dense()
        f = os.path.join(d, f"ex{i}.npz")
        io.save_fsa_npz(f, (em, tr), (em[:2], tr[:2]), [1, 2], [1, 2])
        files.append(f)
    out["write_dense_npz_s_per_example"] = (time.perf_counter() - t0) / N
    out["dense_npz_bytes_per_example"] = os.path.getsize(files[0])
    # ---- dense route
    t = {"load": 0.0, "collate": 0.0, "pack": 0.0, "h2d+fb": 0.0}
    for b0 in range(0, N, BATCH):
        t0 = time.perf_counter(); recs = [io.load_fsa_from_npz(f) for f in files[b0:b0 + BATCH]]; t["load"] += time.perf_counter() - t0
        t0 = time.perf_counter(); col = io.collate(recs, pad=synth.PAD); t["collate"] += time.perf_counter() - t0
        t0 = time.perf_counter(); lat = LatticeBatch.from_dense(col[0], col[1]); t["pack"] += time.perf_counter() - t0
        t0 = time.perf_counter(); r = ops.forward_backward(lat.to(dev), theta); torch.cuda.synchronize(); t["h2d+fb"] += time.perf_counter() - t0
    out["dense_route_ms_per_example"] = {k: v / N * 1e3 for k, v in t.items()}
    out["dense_route_examples_per_s_one_host_process"] = N / sum(t.values())
    # ---- sidecar route: first pass writes the sidecars, second pass is the steady state
    t0 = time.perf_counter(); [io.load_packed(f) for f in files]; out["sidecar_first_pass_ms_per_example"] = (time.perf_counter() - t0) / N * 1e3
    out["sidecar_bytes_per_example"] = os.path.getsize(io.packed_sidecar(files[0]))
    t = {"load_packed": 0.0, "concat": 0.0}
    batches = []
    for b0 in range(0, N, BATCH):
        t0 = time.perf_counter(); ex = [io.load_packed(f) for f in files[b0:b0 + BATCH]]; t["load_packed"] += time.perf_counter() - t0
        t0 = time.perf_counter(); batches.append(io.collate_packed(ex)); t["concat"] += time.perf_counter() - t0
    out["sidecar_route_host_ms_per_example"] = {k: v / N * 1e3 for k, v in t.items()}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    z = [ops.forward_backward(b, theta).logz64.sum() for b in io.DevicePrefetcher(batches, dev)]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out["sidecar_route_prefetch_h2d_fb_ms_per_example"] = dt / N * 1e3
    host = sum(t.values()) / N
    out["sidecar_route_examples_per_s_one_host_process"] = 1.0 / (host + dt / N)
    # the step itself, for scale
    lat = batches[0].to(dev)
    for _ in range(5): ops.forward_backward(lat, theta)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): ops.forward_backward(lat, theta)
    torch.cuda.synchronize()
    out["forward_backward_ms_per_example_resident"] = (time.perf_counter() - t0) / 50 / BATCH * 1e3
    out["host_processes_to_keep_up_with_resident_step"] = {"dense": sum(out["dense_route_ms_per_example"][k] for k in ("load", "collate", "pack")) / out["forward_backward_ms_per_example_resident"],
                                                        "sidecar": host * 1e3 / out["forward_backward_ms_per_example_resident"]}
print(json.dumps(out, indent=1))
