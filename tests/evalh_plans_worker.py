"""worker of test_gpu_evalh.py::test_plan_cache_is_bounded: a process whose plan cache holds TWO programs (H2_EVALH_PLANS_MAX=2,
read once per process) evaluates four programs in turn, twice: every result equals the interpreter's, a program pushed out of
memory comes back from the disk cache (from_cache == 2, not 1) and the one just used is still in memory."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import numpy as np  # noqa: E402
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before HIP initialises (halo2-gpu-specific_amd/__init__.py says why)
import torch  # noqa: E402,F401  (first: the library binds to torch's HIP runtime)

from evalh_cases import random_case  # noqa: E402
from h2util import Oracle  # noqa: E402
from halo2_gpu_specific_amd import evaluation as ev  # noqa: E402


def no_hiprtc():
    """H2_HIPRTC_LIB names a library that is not there: every program runs on the interpreter kernels, same bits"""
    from evalh_cases import oracle_evaluate_h

    oracle = Oracle.get()
    for i in range(2):
        b = ev.Builder().build(**random_case(300 + i, 6 + i, 8 + i, oracle, n_calcs=25))
        assert np.array_equal(ev.evaluate_h(b), oracle_evaluate_h(oracle, b)), i
    assert ev.generated_launches() == 0
    print("no-hiprtc worker ok")


def main():
    if os.environ.get("H2_HIPRTC_LIB"):
        return no_hiprtc()
    assert os.environ["H2_EVALH_PLANS_MAX"] == "2"
    oracle = Oracle.get()
    cases = [random_case(200 + i, 6 + i, 8 + i, oracle, n_calcs=20 + 5 * i) for i in range(4)]
    descs = [ev.Builder().build(**kw) for kw in cases]
    want = [ev.evaluate_h(ev.Builder().build(**kw, flags=ev.EVALH_INTERPRET)) for kw in cases]
    seen = []
    for rnd in range(2):
        for i, b in enumerate(descs):
            got = ev.evaluate_h(b)                       # generates (round 0) or reloads from disk (round 1): never from memory
            assert np.array_equal(got, want[i]), (rnd, i)
            state = ev.prepare(b)["from_cache"]
            assert state == 1, "the program just used must still be loaded"
            seen.append(state)
        # the two most recent programs are in memory, the two before them are not
        assert ev.prepare(descs[3])["from_cache"] == 1 and ev.prepare(descs[2])["from_cache"] == 1
        assert ev.prepare(descs[0])["from_cache"] == 2   # pushed out, found on disk (and now pushes descs[3]... out in turn)
        assert np.array_equal(ev.evaluate_h(descs[0]), want[0])
    print("plans worker ok: %d launches of generated kernels" % ev.generated_launches())


if __name__ == "__main__":
    main()
