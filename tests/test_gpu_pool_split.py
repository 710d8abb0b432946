"""The in-call multi-device splits of the host-slice entry points, EXECUTED on one GPU: with HALO2_PROOFS_N_GPU=4 the pool
has four entries that wrap onto the one visible device (arithmetic.rs:355), so h2_msm_multi runs its four chunks + host fold
(gpu_multiexp_bound, arithmetic.rs:413-440) and h2_evaluate_h_coeff deals its cosets over four leases
(plonk/evaluation.rs:1262-1275) -- both against the oracle; ragged sizes, fewer points than devices, identity parts,
1 / 2 / 4 / 16 cosets.  A child process: the pool size is read once per process."""
import os
import subprocess
import sys

import pytest

from h2util import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_pool_of_four_on_one_device_runs_the_real_splits():
    env = dict(os.environ, HALO2_PROOFS_N_GPU="4")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "pool_split_worker.py"), ROOT], env=env, capture_output=True,
                         text=True, timeout=850)
    assert res.returncode == 0 and "POOL-SPLIT-OK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
    assert res.stdout.count("cosets over the pool ok") == 6
