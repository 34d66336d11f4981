"""Host-side pieces of the mirror classes that need no GPU."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from nfst_amd.samplers import Sampler


@pytest.mark.parametrize("pad", [0, 7])
def test_stripping_pad_matches_reference(golden_dir, pad):
    """Sampler.stripping_pad (samplers.py:162-180), fixture produced by the reference (pad = 0 and a
    pad id different from the dropped mark 0)."""
    d = np.load(os.path.join(golden_dir, "strip.npz"))
    smp = Sampler.__new__(Sampler)
    smp.model = SimpleNamespace(__pad__=pad)
    got = smp.stripping_pad(torch.from_numpy(d[f"pad{pad}_in"])).numpy()
    assert np.array_equal(got, d[f"pad{pad}_out"])
