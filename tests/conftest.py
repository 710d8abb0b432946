import os
import sys

import pytest

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before HIP initialises (halo2-gpu-specific_amd/__init__.py says why)
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


@pytest.fixture(scope="session", autouse=True)
def default_stream_kept_busy():
    """H2_TEST_STALL_ALL=1 (a stream-order check, off by default): a background thread keeps torch's DEFAULT stream busy -- one
    30 ms sleep kernel always queued -- while the GPU tests run.  The prover and the library work on streams of their own, which
    the default stream does not order: a tensor built on the default stream by mistake and consumed on theirs is then not ready,
    and the test that depends on it fails instead of passing by luck (tests/test_gpu_plonk.py runs its multi-rank workers this
    way always)."""
    if os.environ.get("H2_TEST_STALL_ALL") != "1":
        yield
        return
    import threading

    import torch

    if not torch.cuda.is_available():
        yield
        return
    state = {"on": True}

    def busy():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        torch.cuda._sleep(20_000_000)
        b.record()
        b.synchronize()
        cycles = int(0.03 * 20_000_000 / max(a.elapsed_time(b) * 1e-3, 1e-6))
        done = torch.cuda.Event()
        while state["on"]:
            torch.cuda._sleep(cycles)
            done.record()
            done.synchronize()

    t = threading.Thread(target=busy, daemon=True)
    t.start()
    yield
    state["on"] = False
    t.join()
