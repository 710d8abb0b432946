"""Random `Evaluator` programs + committed polynomials for evaluate_h tests (shared by the CPU oracle
pin and the GPU parity test)."""
import random

import numpy as np

from halo2_gpu_specific_amd import evaluation as ev
from h2util import R_MOD, fr_mont, from_mont, to_mont

ZETA = 0x30644E72E131A029048B6E193FD84104CC37A73FEC2BC5E9B8CA0B2D36636F23
DELTA = 0x09226B6E22C6F0CA64EC26AAD4C86E715B5F898E5E963F25870E56BBE533E9A2
ROOT = 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C


def random_case(seed, k, extended_k, oracle, n_calcs=24, with_perm=True, lookup_sets=(1, 3), n_shuffles=2):
    rng = random.Random(seed)
    size = 1 << extended_k
    n_fixed, n_advice, n_instance = 2, 3, 1
    blinding = 5
    rotations = [0, 1, -1, 2, -(blinding + 1)]
    cols = {
        "fixed": [oracle.random_fr(seed * 100 + i, size) for i in range(n_fixed)],
        "advice": [oracle.random_fr(seed * 100 + 10 + i, size) for i in range(n_advice)],
        "instance": [oracle.random_fr(seed * 100 + 20 + i, size) for i in range(n_instance)],
    }
    constants = to_mont([0, 1, 2, R_MOD - 1, rng.randrange(R_MOD), rng.randrange(R_MOD)])

    def rand_vs(n_inter):
        kind = rng.choice([ev.VS_CONSTANT, ev.VS_FIXED, ev.VS_ADVICE, ev.VS_INSTANCE] + [ev.VS_INTERMEDIATE] * (3 if n_inter else 0))
        if kind == ev.VS_CONSTANT:
            return ev.vs(kind, rng.randrange(len(constants)))
        if kind == ev.VS_INTERMEDIATE:
            return ev.vs(kind, rng.randrange(n_inter))
        n = {ev.VS_FIXED: n_fixed, ev.VS_ADVICE: n_advice, ev.VS_INSTANCE: n_instance}[kind]
        return ev.vs(kind, rng.randrange(n), rng.randrange(len(rotations)))

    def rand_calc(n_inter):
        op = rng.choice([ev.CALC_ADD, ev.CALC_SUB, ev.CALC_MUL, ev.CALC_MUL, ev.CALC_NEGATE, ev.CALC_LC_CHALLENGE, ev.CALC_LC_THETA,
                         ev.CALC_ADD_CHALLENGE, ev.CALC_STORE])
        return ev.calc(op, rand_vs(n_inter), rand_vs(n_inter), rng.choice([0, 1]), rng.choice([0, 1, 2, 3, 5]))

    calcs = [rand_calc(i) for i in range(n_calcs)]
    value_parts = [ev.vs(ev.VS_INTERMEDIATE, rng.randrange(n_calcs)) for _ in range(4)] + [rand_vs(n_calcs) for _ in range(2)]
    lookups = [(rand_calc(n_calcs), [rand_calc(n_calcs) for _ in range(s)], [rand_calc(n_calcs) for _ in range(s)]) for s in lookup_sets]
    shuffles = [(rand_calc(n_calcs), rand_calc(n_calcs)) for _ in range(n_shuffles)]
    nz = sum(lookup_sets)
    extra = {
        "l0": oracle.random_fr(seed * 100 + 30, size),
        "l_last": oracle.random_fr(seed * 100 + 31, size),
        "l_active_row": oracle.random_fr(seed * 100 + 32, size),
        "lookup_z": [oracle.random_fr(seed * 100 + 40 + i, size) for i in range(nz)],
        "lookup_m": [oracle.random_fr(seed * 100 + 50 + i, size) for i in range(len(lookup_sets))],
        "shuffle_z": [oracle.random_fr(seed * 100 + 60 + i, size) for i in range(n_shuffles)],
    }
    perm_columns, perm_z, perm_sigma, chunk_len = [], [], [], 2
    if with_perm:
        perm_columns = [(ev.ANY_ADVICE, 0), (ev.ANY_FIXED, 1), (ev.ANY_ADVICE, 2), (ev.ANY_INSTANCE, 0), (ev.ANY_ADVICE, 1)]
        nsets = (len(perm_columns) + chunk_len - 1) // chunk_len
        perm_z = [oracle.random_fr(seed * 100 + 70 + i, size) for i in range(nsets)]
        perm_sigma = [oracle.random_fr(seed * 100 + 80 + i, size) for i in range(len(perm_columns))]
    ch = {n: fr_mont(rng.randrange(R_MOD)) for n in ("y", "beta", "gamma", "theta")}
    ext_omega = pow(ROOT, 1 << (28 - extended_k), R_MOD)
    kw = dict(k=k, extended_k=extended_k, blinding_factors=blinding, chunk_len=chunk_len, constants=constants, rotations=rotations,
              calculations=calcs, value_parts=value_parts, lookups=lookups, shuffles=shuffles, fixed=cols["fixed"],
              advice=cols["advice"], instance=cols["instance"], l0=extra["l0"], l_last=extra["l_last"],
              l_active_row=extra["l_active_row"], perm_z=perm_z, perm_columns=perm_columns, perm_sigma=perm_sigma,
              lookup_z=extra["lookup_z"], lookup_m=extra["lookup_m"], shuffle_z=extra["shuffle_z"], delta=fr_mont(DELTA),
              zeta=fr_mont(ZETA), extended_omega=fr_mont(ext_omega), **ch)
    return kw


def oracle_evaluate_h(oracle, builder):
    import ctypes

    out = np.zeros((1 << builder.desc.extended_k, 4), dtype=np.uint64)
    fn = oracle.lib.oracle_evaluate_h
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    fn.restype = None
    fn(ctypes.byref(builder.desc), out.ctypes.data)
    return out


def python_evaluate_h(kw):
    """Independent big-integer restatement of evaluation.rs:778-1226 for tiny sizes (pins the C oracle)."""
    p = R_MOD
    size = 1 << kw["extended_k"]
    rs = 1 << (kw["extended_k"] - kw["k"])
    I = lambda a: from_mont(a)
    fixed, advice, inst = [I(c) for c in kw["fixed"]], [I(c) for c in kw["advice"]], [I(c) for c in kw["instance"]]
    consts = I(kw["constants"])
    y, beta, gamma, theta = (I(kw[n])[0] for n in ("y", "beta", "gamma", "theta"))
    delta, zeta, w = I(kw["delta"])[0], I(kw["zeta"])[0], I(kw["extended_omega"])[0]
    l0, l_last, lar = I(kw["l0"]), I(kw["l_last"]), I(kw["l_active_row"])
    rots = kw["rotations"]
    values = [0] * size
    lk = [[None] * size for _ in kw["lookups"]]
    sh = [[None] * size for _ in kw["shuffles"]]
    for idx in range(size):
        ri = [(idx + r * rs) % size for r in rots]
        inter = []

        def get(v):
            if v.kind == ev.VS_CONSTANT:
                return consts[v.index]
            if v.kind == ev.VS_INTERMEDIATE:
                return inter[v.index]
            tab = {ev.VS_FIXED: fixed, ev.VS_ADVICE: advice, ev.VS_INSTANCE: inst}[v.kind]
            return tab[v.index][ri[v.rot]]

        def evalc(c):
            a = get(c.a)
            ch = beta if c.challenge == 0 else gamma
            if c.op == ev.CALC_ADD: return (a + get(c.b)) % p
            if c.op == ev.CALC_SUB: return (a - get(c.b)) % p
            if c.op == ev.CALC_MUL: return a * get(c.b) % p
            if c.op == ev.CALC_NEGATE: return (-a) % p
            if c.op == ev.CALC_LC_CHALLENGE: return (a + (pow(ch, c.power, p) if c.power > 1 else ch)) * get(c.b) % p
            if c.op == ev.CALC_LC_THETA: return (a * theta + get(c.b)) % p
            if c.op == ev.CALC_ADD_CHALLENGE: return (a + ch) % p
            return a

        for c in kw["calculations"]:
            inter.append(evalc(c))
        v = 0
        for part in kw["value_parts"]:
            v = (v * y + get(part)) % p
        values[idx] = v
        for t, (table, prods, sums) in enumerate(kw["lookups"]):
            lk[t][idx] = (evalc(table), [evalc(c) for c in prods], [evalc(c) for c in sums])
        for i, (a, b) in enumerate(kw["shuffles"]):
            sh[i][idx] = (evalc(a), evalc(b))
    last_rot = -(kw["blinding_factors"] + 1)
    pz, ps = [I(z) for z in kw["perm_z"]], [I(s) for s in kw["perm_sigma"]]
    if pz:
        cl = kw["chunk_len"]
        colv = [{ev.ANY_ADVICE: advice, ev.ANY_FIXED: fixed, ev.ANY_INSTANCE: inst}[t][i] for t, i in kw["perm_columns"]]
        for idx in range(size):
            rn, rl = (idx + rs) % size, (idx + last_rot * rs) % size
            v = values[idx]
            v = (v * y + (1 - pz[0][idx]) * l0[idx]) % p
            v = (v * y + (pz[-1][idx] ** 2 - pz[-1][idx]) * l_last[idx]) % p
            for s in range(1, len(pz)):
                v = (v * y + (pz[s][idx] - pz[s - 1][rl]) * l0[idx]) % p
            cur = beta * zeta % p * pow(w, idx, p) % p
            for s in range(len(pz)):
                left, right = pz[s][rn], pz[s][idx]
                for j in range(s * cl, min((s + 1) * cl, len(colv))):
                    left = left * (colv[j][idx] + beta * ps[j][idx] + gamma) % p
                for j in range(s * cl, min((s + 1) * cl, len(colv))):
                    right = right * (colv[j][idx] + cur + gamma) % p
                    cur = cur * delta % p
                v = (v * y + (left - right) * lar[idx]) % p
            values[idx] = v
    lz, lm = [I(z) for z in kw["lookup_z"]], [I(m) for m in kw["lookup_m"]]
    zo = 0
    for t, (table, prods, sums) in enumerate(kw["lookups"]):
        n = len(prods)
        zs = lz[zo:zo + n]
        zo += n
        for idx in range(size):
            rn, rl = (idx + rs) % size, (idx + last_rot * rs) % size
            tb, pr, sm = lk[t][idx]
            v = values[idx]
            v = (v * y + zs[0][idx] * l0[idx]) % p
            v = (v * y + zs[n - 1][idx] * l_last[idx]) % p
            v = (v * y + (((zs[0][rn] - zs[0][idx]) * tb + lm[t][idx]) * pr[0] - tb * sm[0]) * lar[idx]) % p
            for i in range(1, n):
                v = (v * y + (zs[i][idx] - zs[i - 1][rl]) * l0[idx]) % p
            for i in range(1, n):
                v = (v * y + ((zs[i][rn] - zs[i][idx]) * pr[i] - sm[i]) * lar[idx]) % p
            values[idx] = v
    sz = [I(z) for z in kw["shuffle_z"]]
    for i in range(len(kw["shuffles"])):
        for idx in range(size):
            rn = (idx + rs) % size
            a, b = sh[i][idx]
            z = sz[i]
            v = values[idx]
            v = (v * y + (1 - z[idx]) * l0[idx]) % p
            v = (v * y + (z[idx] ** 2 - z[idx]) * l_last[idx]) % p
            v = (v * y + (z[rn] * b - z[idx] * a) * lar[idx]) % p
            values[idx] = v
    return values
