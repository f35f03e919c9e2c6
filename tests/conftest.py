import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def cfg2_seeded_inputs(g, seed=21, Ta=400, Tt=128, d=768):
    """inputs of tests/golden/cfg2_seeded.npz (the headline shape, B=2) and of its cfg4 / cfg5 siblings: regenerated from the
    generator seed make_golden.py used, verified against the stored probes -- the fixtures hold no inputs"""
    B = 2
    gen = torch.Generator().manual_seed(seed)
    h_a = torch.randn(B, Ta, d, generator=gen)
    h_t = torch.randn(B, Tt, d, generator=gen)
    probe = torch.cat([h_a[0, 0, :8], h_t[1, -1, -8:], h_a.sum().reshape(1), h_t.sum().reshape(1)])
    assert torch.allclose(probe, g["probe"], rtol=1e-6, atol=1e-4), "the seeded generator no longer reproduces the fixture's inputs"
    return h_a, h_t, g["mask_a"], g["mask_t"]
