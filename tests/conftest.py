import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    # torch bundles its own HIP runtime: when it is used in the same process as libhalo2_hip.so it has to
    # initialise first, so that both bind to one runtime instance (the loader reuses the loaded SONAME)
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:  # noqa: BLE001 - torch is optional plumbing for the tests
        pass


@pytest.fixture(scope="session")
def oracle():
    from h2util import Oracle

    return Oracle.get()
