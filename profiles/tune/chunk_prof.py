"""For rocprofv3 --kernel-trace --stats: the SNIPS-shaped batch of 64 through the chunked flavour, 200 launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nfst_amd import ops, synth
from nfst_amd.lattice import LatticeBatch
dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lat = LatticeBatch.from_synth(synth.snips_shaped_batch(B, vocab=250), device=dev)
assert lat.chunks is not None
theta = torch.from_numpy(synth.label_scores(1, 250, mean=-1.5, std=0.8)).to(dev)
out = None
for _ in range(200):
    out = ops.forward_backward(lat, theta, out=out)
torch.cuda.synchronize()
