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


@pytest.mark.parametrize("n,H", [(5, 3), (300, 8), (256 * 64, 16), (256 * 64 + 37, 8), (50000, 5)])
def test_gram_of_tall_operands(n, H):
    """ops._gram (the host-side product of the neuralised beta's backward, dL/dWh = gamma^T beta_hat) is a^T b
    whatever the row count: chunked as a batched GEMM above 16k rows, the remainder added separately."""
    from nfst_amd.ops import _gram
    g = torch.Generator().manual_seed(n + H)
    a = torch.randn(n, H, generator=g, dtype=torch.float64)
    b = torch.randn(n, H, generator=g, dtype=torch.float64)
    assert torch.allclose(_gram(a, b), a.t() @ b, rtol=1e-12, atol=1e-9)
