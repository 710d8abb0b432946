"""Product-side descriptions (halo2-gpu-specific_amd.circuit.ConstraintSystem) of the test circuits whose big-integer
twins live in ref_plonk.py; shared by the CPU and GPU proof tests and by ref_plonk's verifying-key digest."""
from halo2_gpu_specific_amd import circuit as hc
from halo2_gpu_specific_amd import circuits


def mini_plonk_cs():
    return circuits.mini_plonk()


def wide_cs(quads):
    return circuits.wide(quads)


def range_check_cs(vmin, vmax, step):
    return circuits.range_check(vmin, vmax, step)


def lookup_api_cs():
    return circuits.lookup_api()


def shuffle_api_group_cs():
    return circuits.shuffle_api_group()


def lookup_api_set_cs():
    return circuits.lookup_api_set()


def shuffle_api_cs():
    return circuits.shuffle_api()


def shuffle_gates_cs(width=4, theta=111, beta=222):
    return circuits.shuffle_gates(width, theta, beta)


def rot_gate_cs():
    """the product-side description of ref_plonk.RotGate"""
    cs = hc.ConstraintSystem("rot-gate")
    a, b, c = cs.advice_column(), cs.advice_column(), cs.advice_column()
    s0, s1 = cs.fixed_column(), cs.fixed_column()
    for col in (a, b, c):
        cs.enable_equality(col)
    q0, q1 = cs.query_fixed(s0), cs.query_fixed(s1)
    cs.enable_equality(s1)
    cs.set_minimum_degree(4)
    cs.create_gate("sum", [q0 * (cs.query_advice(a) + cs.query_advice(b) - cs.query_advice(c))])
    cs.create_gate("step", [q1 * (cs.query_advice(a, 1) - cs.query_advice(c)) * (cs.query_advice(b, -1) + q0)])
    return cs


def lookup_shuffle_cs():
    """the product-side description of ref_plonk.LookupShuffle"""
    cs = hc.ConstraintSystem("lookup-shuffle")
    adv = [cs.advice_column() for _ in range(12)]
    fx = [cs.fixed_column() for _ in range(5)]
    inst = cs.instance_column()
    cs.enable_equality(adv[11])
    cs.enable_equality(inst)
    q, qi = cs.query_fixed(fx[0]), cs.query_fixed(fx[4])
    a, b, c, d, e, g, h, g2, h2, p, p2 = (cs.query_advice(adv[i]) for i in range(11))
    w, pub = cs.query_advice(adv[11]), cs.query_instance(inst)
    t0, t1, u = cs.query_fixed(fx[1]), cs.query_fixed(fx[2]), cs.query_fixed(fx[3])
    cs.create_gate("square", [q * (a * a + 1 - b)])
    cs.create_gate("public", [qi * (w - pub)])
    cs.lookup_any("pairs", [t0, t1], [[[q * a, q * b], [c, d]], [[q * c, q * d]]])
    cs.lookup_any("single", [u], [[[e]]])
    cs.shuffle_group([("gh", [g, h], [g2, h2]), ("p", [p], [p2])])
    cs.set_minimum_degree(6)
    return cs
