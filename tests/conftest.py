import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Golden .npz (written by tests/golden/make_golden.py) -> dict of torch tensors / numpy scalars."""
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        out = {}
        for k in z.files:
            a = z[k]
            if a.dtype.kind in "fiub" and a.ndim > 0:
                out[k] = torch.from_numpy(a.copy())
            else:
                out[k] = a
        return out


def master_params(n_blocks=3):
    p = load_golden("master_params.npz")
    keep = {}
    for k, v in p.items():
        if k.startswith("upsample_blocks.") and int(k.split(".")[1]) >= n_blocks:
            continue
        keep[k] = v
    return keep


@pytest.fixture(scope="session")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
