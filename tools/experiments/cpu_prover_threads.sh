#!/bin/bash
# CPU create_proof (tests/oracle_prover.py) against the thread count, k = 20 mini-PLONK: what the CPU baseline should run on
cd /root/repo/tests
for t in 16 32 64 128 256; do
H2_ORACLE_THREADS=$t python - <<PY
import sys, time; sys.path.insert(0, '..')
import oracle_prover as op
from h2util import Oracle
from halo2_gpu_specific_amd import circuits, prover
from halo2_gpu_specific_amd.rng import ProverRng
o = Oracle.get(); k = 20; n = 1 << k
D = op.OracleDevice()
params = prover.Params(D, k, o.random_g1(1, n), o.random_g1(2, n), tables=False)
adv, fixed, copies = circuits.mini_plonk_synthesize(k)
t0 = time.time(); pk = op.keygen(D, params, circuits.mini_plonk(), fixed, copies); t1 = time.time()
tm = {}
prover.create_proof_ext(D, params, pk, adv, ProverRng(1), False, timings=tm); t2 = time.time()
print("threads", D.L.threads, "keygen %.2f prove %.2f" % (t1 - t0, t2 - t1), {a: round(b, 2) for a, b in tm.items()})
PY
done
