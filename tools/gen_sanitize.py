"""The evaluate_h generator (csrc/evalh_gen.cpp, host C++) under AddressSanitizer + UBSan on the CPU: every product circuit, the
gate sets of the probe and N random circuits of tools/prover_fuzz.py are generated and compiled (hipRTC cross-compiles gfx950
without a GPU) with several option sets, twice each (fresh, then from the disk cache).  Driven by tools/gen_sanitize.sh, which
builds the instrumented library into a scratch directory and points H2_LIB at it."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import halo2_gpu_specific_amd as h2  # noqa: E402
from halo2_gpu_specific_amd import circuits, evaluation as ev, prover  # noqa: E402

import prover_fuzz  # noqa: E402

OPTION_SETS = [{}, {"H2_JIT_FACTOR": "0"}, {"H2_JIT_STAGE_PRODUCTS": "24"}, {"H2_JIT_MUL2": "0", "H2_JIT_LIVE": "8"},
               {"H2_JIT_LDS_ARGS": "1"}, {"H2_JIT_MIN_GROUP": "1", "H2_JIT_GAP": "2"}]


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    h2.lib()
    cases = [("mini-PLONK", circuits.mini_plonk(), 6), ("wide-16", circuits.wide(16), 6), ("wide-64", circuits.wide(64), 6)]
    for seed in range(count):
        cs, k, *_ = prover_fuzz.random_case(1000 + seed, satisfiable=bool(seed & 1))
        cases.append(("fuzz-%d" % (1000 + seed), cs, k))
    t0, done = time.time(), 0
    for name, cs, k in cases:
        ext = k + max(1, (cs.degree() - 1 - 1).bit_length())
        desc = prover.program_descriptor(cs, k, ext)
        for i, opts in enumerate(OPTION_SETS if name.startswith(("mini", "wide")) else OPTION_SETS[:1 + sum(map(ord, name)) % 3]):
            for key in [k_ for k_ in os.environ if k_.startswith("H2_JIT_") and k_ != "H2_JIT_CACHE"]:
                del os.environ[key]
            os.environ.update(opts)
            first = ev.compile_only(desc)
            again = ev.compile_only(desc)
            assert first["products_per_row"] == again["products_per_row"] and again["from_cache"] in (1, 2), (name, opts, first, again)
            src = ev.generated_source(desc)
            assert "h2_evalh_gen" in src
            done += 1
        print("%-12s degree %d: %d stages, %d products per row, %d registers" % (
            name, cs.degree(), first["stages"], first["products_per_row"], first["max_registers"]), flush=True)
    print("gen_sanitize: %d programs generated and compiled twice in %.0f s, no sanitizer report" % (done, time.time() - t0))


if __name__ == "__main__":
    main()
