"""Random satisfied circuits (tools/prover_fuzz.py) proved by 2 / 3 / 4 / 8 gloo ranks sharing cuda:0 -- the multi-rank data flow
of DESIGN.md section 6: range-split MSMs, cosets and coset rank groups, row-range products / lookups / evaluations / multiopen,
columns dealt for the inverse transforms -- against the single-device proof bytes.  The workers are the ones of
tests/test_gpu_plonk.py (torch's default stream kept busy, uninitialised vectors poisoned).

usage: python tools/multirank_fuzz.py [seconds] [first seed]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    import test_gpu_plonk as T
    from halo2_gpu_specific_amd import prover
    from halo2_gpu_specific_amd.rng import ProverRng

    D = prover.Device(0)
    t0, cases, skipped = time.time(), 0, 0
    tmp = tempfile.mkdtemp()
    while time.time() - t0 < seconds:
        world = (2, 4, 8, 3, 8, 4)[seed % 6]
        k = 8 + seed % 3
        which = "fuzz:%d" % seed
        try:
            cs, adv, fixed, copies, inst = T._multi_rank_case(which, k)
        except AssertionError:                       # the drawn circuit needs more rows than 2^k
            seed += 1
            skipped += 1
            continue
        params = prover.Params.unsafe_setup(D, k, T.S_TRAPDOOR)
        pk = prover.keygen(D, params, cs, fixed, copies)
        try:
            want = [prover.create_proof_ext(D, params, pk, adv, ProverRng(9), gwc, instances=inst) for gwc in (False, True)]
        except ValueError as e:                      # a quotient piece that is identically zero commits to the identity, which the
            print("seed %d: %s (the reference fails the same way: transcript.rs:203-209) -- skipped" % (seed, e))   # transcript refuses
            seed += 1
            skipped += 1
            continue
        script = os.path.join(tmp, "worker.py")
        open(script, "w").write(T._WORKER % (ROOT, os.path.join(ROOT, "tests"), which, k, T.S_TRAPDOOR))
        res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
                              "--master-addr", "127.0.0.1", "--master-port", str(T._free_port()), script],
                             capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, H2_TEST_BACKEND="gloo", H2_POISON_EMPTY="1"))
        ok = res.returncode == 0
        for tag, proof in (("PROOF", want[0]), ("GWC", want[1])):
            got = [l.split()[1] for l in res.stdout.splitlines() if l.startswith(tag + " ")]
            ok = ok and len(got) == world and all(T._same_proof(h, proof) for h in got)
        print("seed %d: %d ranks, k = %d, degree %d, %d lookups, %d shuffle groups, cosets %d: %s" % (
            seed, world, k, cs.degree(), len(cs.lookups), len(cs.shuffles), pk.domain.quotient_poly_degree, "ok" if ok else "MISMATCH"),
            flush=True)
        if not ok:
            print(res.stdout[-2000:] + res.stderr[-3000:])
            sys.exit(1)
        cases += 1
        seed += 1
    print("multirank_fuzz: %d circuits (%d draws skipped), every rank of every world emitted the single-device bytes" % (cases, skipped))


main()
