"""Phase stamps of k_pack_lattice (profiling build: python -m nfst_amd.build --variant pkstamps -DNFST_PK_STAMPS; run with
NFST_TUNING=1 NFST_LIB=nfst_amd/lib/variants/libnfst_hip_pkstamps.so).  Workgroup 0; 100 MHz wall clock."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from nfst_amd import synth, _lib
from nfst_amd.lattice import LatticeBatch
dev = torch.device("cuda")
for name, lats, V in (("baseline256", synth.bench_batch(256), 256), ("snips64", synth.snips_shaped_batch(64), 250)):
    n_rows, arc_off, src, label, dst, w = synth.batch_arcs(lats)
    for rep in range(2):
        lat = LatticeBatch.from_arcs_device(n_rows, arc_off, src, label, dst, V, device=dev)
        torch.cuda.synchronize()
    buf = (C.c_ulonglong * 32)()
    _lib.lib._handle  # noqa
    fn = _lib.lib.nfst_debug_pk_stamps
    fn.argtypes = [C.c_void_p]; fn.restype = C.c_int
    assert fn(buf) == 0
    st = np.array(list(buf), dtype=np.int64)
    names = ["entry", "checked", "relaxed", "degrees", "lists", "in-rank", "level orders", "(emit) canonical", "layouts"]
    for base, tag in ((0, "plan"), (16, "emit")):
        t = st[base:base + 9]
        print(name, tag, "lattice 0 rows", lats[0].n_rows, "arcs", lats[0].n_arcs, {names[k]: round((t[k] - t[k - 1]) / 100.0, 1) for k in range(1, 9) if t[k] and t[k - 1]}, "us; total", (t[8] - t[0]) / 100.0)
