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


# ---- the CPU side of the k = 24 twin runs UNDER the other GPU tests (tests/cpu_prover_worker.py) --------------------------
K24_TWIN = "test_full_size_proof_bytes_equal_cpu_proof_bytes[24]"
K24_SEED = 22


@pytest.hookimpl(trylast=True)
def pytest_collection_modifyitems(config, items):
    """when the k = 24 CPU twin is among the selected tests on a GPU box: its CPU half (3 minutes of host cores) starts now, in
    a process of its own, and the test moves to the END of the session -- the wait that used to sit in front of every later
    test overlaps with them.  H2_TEST_K24_WORKER=0: everything inline, as before."""
    twin = [it for it in items if it.nodeid.endswith(K24_TWIN)]
    if not twin or os.environ.get("H2_TEST_K24_WORKER") == "0" or len(items) < 20:
        return
    try:
        import torch

        if not torch.cuda.is_available():
            return
    except Exception:  # noqa: BLE001
        return
    import subprocess
    import tempfile

    items[:] = [it for it in items if it not in twin] + twin
    out = os.path.join(tempfile.mkdtemp(prefix="h2_k24_twin_"), "cpu_proof.json")
    env = dict(os.environ)
    env.setdefault("H2_ORACLE_THREADS", "10")        # of the 16 CPUs the GPU box grants: the tests in front keep the rest
    log = open(out + ".log", "w")
    proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "cpu_prover_worker.py"), "24", str(K24_SEED), "0", out],
                            env=env, stdout=log, stderr=subprocess.STDOUT)
    config._h2_k24_worker = {"proc": proc, "out": out, "log": out + ".log"}


def pytest_sessionfinish(session, exitstatus):
    w = getattr(session.config, "_h2_k24_worker", None)
    if w and w["proc"].poll() is None:
        w["proc"].kill()


@pytest.fixture
def k24_cpu_worker(request):
    """the running worker's handle ({"proc", "out", "log"}) or None"""
    return getattr(request.config, "_h2_k24_worker", None)
