import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """tests/golden/lsd_golden.npz -- vectors generated from the reference's own CPU code
    (tests/golden/make_golden.py); loaded without pickle."""
    path = os.path.join(ROOT, "tests", "golden", "lsd_golden.npz")
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle

    oracle.lib()   # builds liboracle.so on first use if absent
    return oracle


@pytest.fixture(scope="session")
def gpu():
    """The product on cuda:0; fails (not skips) when the HIP library cannot be used."""
    import torch

    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    import lsdradixsort_amd as lsd

    assert lsd.lib().lsdsort_device_count() >= 1, "liblsdsort.so sees no gfx950 device"
    torch.cuda.set_device(0)
    return lsd
