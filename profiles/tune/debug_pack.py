"""Device packer, planning pass only, on one lattice of the test corpus: status, meta and the workspace's depth / height rows."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from nfst_amd import _lib, synth
from nfst_amd._lib import lib
from nfst_amd.lattice import LatticeBatch
import test_gpu_pack as T
name, idx = sys.argv[1], int(sys.argv[2])
l = T.corpus()[name][idx]
dev = torch.device("cuda:0")
n_rows, arc_off, src, label, dst, w = synth.batch_arcs([l])
B = 1
row_off = np.array([0, l.n_rows], np.int64)
t = lambda x, dt: torch.from_numpy(np.ascontiguousarray(x, dtype=dt)).to(dev)
n_rows_d, row_off_d, arc_off_d, s_d, l_d, d_d = t(n_rows, np.int32), t(row_off, np.int64), t(arc_off, np.int64), t(src, np.int32), t(label, np.int32), t(dst, np.int32)
arcs = _lib.ArcsDevice(n_rows_d.data_ptr(), row_off_d.data_ptr(), arc_off_d.data_ptr(), s_d.data_ptr(), l_d.data_ptr(), d_d.data_ptr(), None,
                       int(row_off[-1]), int(arc_off[-1]), B, int(l.vocab))
ws_bytes = int(lib.nfst_pack_device_ws_bytes(B, int(row_off[-1]), int(arc_off[-1])))
ws = torch.full((ws_bytes // 4 + 4,), -77, dtype=torch.int32, device=dev)
plan = torch.zeros(B * 18, dtype=torch.int32, device=dev)
opts = _lib.PackOpts(0, 0, 0, 0)
rc = lib.nfst_pack_device_plan(C.byref(arcs), C.byref(opts), ws.data_ptr(), ws_bytes, plan.data_ptr(), plan[16:].data_ptr(), plan[17:].data_ptr(),
                               torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
p = plan.cpu().numpy()
print("rc", rc, "status", p[16], "scratch", p[17], "meta", p[:16])
host = LatticeBatch.from_synth([l])
print("host meta", host.meta_host[0])
n = l.n_rows
wsn = ws.cpu().numpy()
dep, hei = wsn[:n], wsn[n + 2:2 * n + 2]
print("dep", dep[:40], "max", dep.max(), "min", dep.min())
print("hei", hei[:40], "max", hei.max())
