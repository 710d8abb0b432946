import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from h2util import Oracle
from evalh_cases import random_case, oracle_evaluate_h
from halo2_gpu_specific_amd import evaluation as ev
oracle = Oracle.get()
kw = random_case(1, 2, 3, oracle, n_calcs=40)
want = oracle_evaluate_h(oracle, ev.Builder().build(**kw))
print("interpreter...", flush=True)
got = ev.evaluate_h(ev.Builder().build(**kw, flags=ev.EVALH_INTERPRET))
print("interpreter ok", np.array_equal(got, want), flush=True)
b = ev.Builder().build(**kw)
print(ev.prepare(b), flush=True)
got = ev.evaluate_h(b)
print("generated", np.array_equal(got, want), flush=True)
