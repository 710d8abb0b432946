"""The slice of `plonk::circuit` the prover's hot path needs on the host: column allocation, queries, custom gates,
the equality (permutation) argument, and the flattening of gate expressions into the `Calculation` program that
h2_dev_evaluate_h interprets.

  ConstraintSystem   plonk/circuit.rs:1283-1956  (advice_column, fixed_column, enable_equality, create_gate,
                                                  query_*_index, degree :1881-1914, blinding_factors :1919-1944)
  Expression         plonk/circuit.rs:609-1060
  GraphEvaluator     plonk/evaluation.rs:298-560  (add_expression / add_calculation / add_constant / add_rotation)

  logup / shuffle    plonk/logup.rs:11-50, plonk/shuffle.rs:8-54 -- arguments are registered with their input sets /
                     groups given explicitly (the result of the reference's chunking passes)

Circuits are custom gates + copy constraints + logup lookups + shuffle groups over advice / fixed / instance columns.
"""
from . import evaluation as ev

R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


class Expression:
    def __add__(self, o):
        return Sum(self, _wrap(o))

    def __sub__(self, o):
        return Sum(self, Negated(_wrap(o)))

    def __mul__(self, o):
        if isinstance(o, int):
            return Scaled(self, o % R_MOD)
        return Product(self, o)

    __radd__ = __add__
    __rmul__ = __mul__

    def __neg__(self):
        return Negated(self)


def _wrap(o):
    return Constant(o % R_MOD) if isinstance(o, int) else o


class Constant(Expression):
    def __init__(self, v):
        self.v = v

    def degree(self):
        return 0

    def identifier(self):
        return "0x%064x" % self.v


class Query(Expression):
    kind = None

    def __init__(self, column, rotation):
        self.column, self.rotation = column, rotation

    def degree(self):
        return 1

    def identifier(self):
        """Expression::identifier (plonk/circuit.rs:805-837): keys the lookup tracer by table"""
        return "%s[%d][%d]" % (self.name, self.column, self.rotation)


class Fixed(Query):
    kind, name = ev.VS_FIXED, "fixed"


class Advice(Query):
    kind, name = ev.VS_ADVICE, "advice"


class Instance(Query):
    kind, name = ev.VS_INSTANCE, "instance"


class Negated(Expression):
    def __init__(self, e):
        self.e = e

    def degree(self):
        return self.e.degree()

    def identifier(self):
        return "(-%s)" % self.e.identifier()


class Sum(Expression):
    def __init__(self, a, b):
        self.a, self.b = a, b

    def degree(self):
        return max(self.a.degree(), self.b.degree())

    def identifier(self):
        return "(%s+%s)" % (self.a.identifier(), self.b.identifier())


class Product(Expression):
    def __init__(self, a, b):
        self.a, self.b = a, b

    def degree(self):
        return self.a.degree() + self.b.degree()

    def identifier(self):
        return "(%s*%s)" % (self.a.identifier(), self.b.identifier())


class Scaled(Expression):
    def __init__(self, e, c):
        self.e, self.c = e, c

    def degree(self):
        return self.e.degree()

    def identifier(self):
        return "%s*0x%064x" % (self.e.identifier(), self.c)


class ConstraintSystem:
    def __init__(self, name="circuit"):
        self.name = name
        self.num_advice = self.num_fixed = self.num_instance = 0
        self.advice_queries, self.fixed_queries, self.instance_queries = [], [], []
        self.num_advice_queries = []
        self.gates = []              # (name, [Expression])
        self.perm_columns = []       # ("advice" | "fixed" | "instance", index)
        self.lookups = []            # (name, [table Expression], [[[input Expression]]]): logup::Argument (plonk/logup.rs:11-16)
        self.shuffles = []           # groups of (name, [input Expression], [shuffle Expression]): shuffle::Argument (plonk/shuffle.rs:8-22)
        self.minimum_degree = None
        self.lookup_tracer = {}      # table identifier -> (name, [table Expression], [(name, [input Expression])])
        self.shuffle_tracer = []     # (name, [input Expression], [shuffle Expression])
        self.range_checks = []       # (origin advice index, sort advice index, min, max, step): range_check::Argument

    # -- columns ------------------------------------------------------------------------------------------
    def advice_column(self):
        self.num_advice += 1
        self.num_advice_queries.append(0)
        return ("advice", self.num_advice - 1)

    def advice_column_range(self, l_0, l_active, l_last_active, vmin, vmax, step):
        """`advice_column_range` (plonk/circuit.rs:1769-1826; plonk/range_check.rs): an advice column whose values lie in
        {vmin, ..., vmax}.  Allocates the column (`origin`) and a companion (`sort`) that holds the same multiset in
        ascending order, a gate -- sort starts at vmin (l_0), ends at vmax (l_last_active), and neighbouring rows differ
        by 0 .. step -- and a shuffle between the two; the prover completes the witness (prover.complete_range_check_witness:
        every value of the range is planted in the unused cells, `sort` is the counting sort).  Returns `origin`."""
        assert step != 0 and vmin <= vmax
        origin, sort = self.advice_column(), self.advice_column()
        q0 = self.query_fixed(l_0)
        first = q0 * (Constant(vmin % R_MOD) - self.query_advice(sort))
        ql = self.query_fixed(l_last_active)
        last = ql * (Constant(vmax % R_MOD) - self.query_advice(sort))
        acc = None
        for i in range(step + 1):
            e = self.query_advice(sort, 1) - self.query_advice(sort) - Constant((step - i) % R_MOD)
            acc = e if acc is None else acc * e
        self.create_gate("range check", [first, last, (self.query_fixed(l_active) - self.query_fixed(l_last_active)) * acc])
        self.shuffle("range check col", [(self.query_advice(origin), self.query_advice(sort))])
        self.range_checks.append((origin[1], sort[1], vmin, vmax, step))
        return origin

    def fixed_column(self):
        self.num_fixed += 1
        return ("fixed", self.num_fixed - 1)

    def instance_column(self):
        self.num_instance += 1
        return ("instance", self.num_instance - 1)

    # -- queries (circuit.rs:1478-1560) -----------------------------------------------------------------------
    def _query(self, column, rotation):
        kind, idx = column
        lst = {"advice": self.advice_queries, "fixed": self.fixed_queries, "instance": self.instance_queries}[kind]
        if (idx, rotation) not in lst:
            lst.append((idx, rotation))
            if kind == "advice":
                self.num_advice_queries[idx] += 1
        return lst.index((idx, rotation))

    def query_advice(self, column, rotation=0):
        assert column[0] == "advice"
        self._query(column, rotation)
        return Advice(column[1], rotation)

    def query_fixed(self, column, rotation=0):
        assert column[0] == "fixed"
        self._query(column, rotation)
        return Fixed(column[1], rotation)

    def query_instance(self, column, rotation=0):
        assert column[0] == "instance"
        self._query(column, rotation)
        return Instance(column[1], rotation)

    def get_any_query_index(self, column, rotation=0):
        kind, idx = column
        lst = {"advice": self.advice_queries, "fixed": self.fixed_queries, "instance": self.instance_queries}[kind]
        return lst.index((idx, rotation))

    def enable_equality(self, column):
        """circuit.rs:1437-1441: queries the column at the current rotation and adds it to the permutation"""
        self._query(column, 0)
        if column not in self.perm_columns:
            self.perm_columns.append(column)

    def create_gate(self, name, polys):
        assert polys, "Gates must contain at least one constraint."
        self.gates.append((name, list(polys)))

    def set_minimum_degree(self, d):
        self.minimum_degree = d

    def lookup_any(self, name, table_expressions, input_expressions_sets):
        """One logup argument with its input sets given explicitly (what `chunk_lookups` produces from the traced
        `lookup_any` calls, plonk/logup.rs:73-153): set 0 shares its polynomial with the table term."""
        for st in input_expressions_sets:
            for inputs in st:
                assert len(inputs) == len(table_expressions)
        self.lookups.append((name, list(table_expressions), [[list(i) for i in st] for st in input_expressions_sets]))

    # -- the traced front end and its chunking passes -----------------------------------------------------------
    def lookup(self, name, pairs):
        """`lookup_any` (plonk/circuit.rs:1377-1406): pairs = [(input Expression, table Expression)]; lookups into the
        same table expressions are collected under one tracer entry"""
        inputs, table = [i for i, _ in pairs], [t for _, t in pairs]
        ident = "".join(t.identifier() for t in table)
        if ident in self.lookup_tracer:
            self.lookup_tracer[ident][2].append((name, inputs))
        else:
            self.lookup_tracer[ident] = (name, table, [(name, inputs)])
        return len(self.lookup_tracer) - 1

    def chunk_lookups(self):
        """circuit.rs:1411-1424 + ArgumentTracer::chunks (plonk/logup.rs:73-153): pack the inputs of every table into
        sets whose polynomial fits the constraint system's degree; set 0 also carries the table term"""
        if not self.lookup_tracer:
            return self
        degree = self.degree()
        assert degree > 2
        max_degree = degree - 2
        deg = lambda exprs: max(e.degree() for e in exprs)  # noqa: E731
        self.lookups = []
        for ident in sorted(self.lookup_tracer):              # BTreeMap<String, _> iteration order
            name, table, traced = self.lookup_tracer[ident]
            first, extra = [traced[0][1]], []
            for _, inputs in traced[1:]:
                d = deg(inputs)
                if deg(table) + sum(deg(i) for i in first) + d <= max_degree:
                    first.append(inputs)
                    continue
                for st in extra:
                    if sum(deg(i) for i in st) + d <= max_degree:
                        st.append(inputs)
                        break
                else:
                    extra.append([inputs])
            self.lookups.append((name, list(table), [first] + extra))
        return self

    def shuffle(self, name, pairs):
        """circuit.rs:1430-1442: pairs = [(input Expression, shuffle Expression)]"""
        self.shuffle_tracer.append((name, [i for i, _ in pairs], [s_ for _, s_ in pairs]))
        return len(self.shuffle_tracer) - 1

    def chunk_shuffles(self):
        """circuit.rs:1445-1451 + shuffle::chunk (plonk/shuffle.rs:57-103): first-fit grouping by summed degree"""
        if not self.shuffle_tracer:
            return self
        degree = self.degree()
        assert degree > 2
        max_degree = degree - 2
        udeg = lambda u: max([1] + [e.degree() for e in u[1] + u[2]])  # noqa: E731
        groups = [[self.shuffle_tracer[0]]]
        for unit in self.shuffle_tracer[1:]:
            for group in groups:
                if sum(udeg(u) for u in group) + udeg(unit) <= max_degree:
                    group.append(unit)
                    break
            else:
                groups.append([unit])
        self.shuffles = groups
        return self

    def shuffle_group(self, units):
        """units: [(name, [input Expression], [shuffle Expression])] sharing one product polynomial"""
        for _, inp, shf in units:
            assert len(inp) == len(shf)
        self.shuffles.append([(nm, list(i), list(sh)) for nm, i, sh in units])

    # -- derived quantities -------------------------------------------------------------------------------
    def degree(self):
        d = 3  # permutation::Argument::required_degree (plonk/permutation.rs:42-69)
        for _, polys in self.gates:
            for p in polys:
                d = max(d, p.degree())
        for _, table, sets in self.lookups:      # logup::Argument::required_degree (plonk/logup.rs:31-50)
            tdeg = max([1] + [e.degree() for e in table])
            ideg = max([1] + [e.degree() for st in sets for inputs in st for e in inputs])
            d = max(d, 4, 2 + ideg + tdeg)
        for _, table, traced in self.lookup_tracer.values():   # ArgumentTracer::required_degree (plonk/logup.rs:155-176)
            tdeg = max([1] + [e.degree() for e in table])
            ideg = max([1] + [e.degree() for _, inputs in traced for e in inputs])
            d = max(d, 4, 2 + ideg + tdeg)
        for group in list(self.shuffles) + [[u] for u in self.shuffle_tracer]:   # ArgumentUnit::required_degree (shuffle.rs:43-54)
            for _, inp, shf in group:
                d = max(d, 2 + max([1] + [e.degree() for e in inp + shf]))
        d = max(d, self.minimum_degree or 1)
        # what the chunking of the reference guarantees (logup.rs:88-150, shuffle.rs:57-80): every argument polynomial fits
        for _, table, sets in self.lookups:
            tdeg = max([1] + [e.degree() for e in table])
            for si, st in enumerate(sets):
                total = 2 + (tdeg if si == 0 else 0) + sum(max([1] + [e.degree() for e in inputs]) for inputs in st)
                assert total <= d, "lookup input set %d needs degree %d > %d" % (si, total, d)
        for group in self.shuffles:
            total = 2 + sum(max([1] + [e.degree() for e in inp + shf]) for _, inp, shf in group)
            assert total <= d, "shuffle group needs degree %d > %d" % (total, d)
        return d

    def blinding_factors(self):
        factors = max(self.num_advice_queries) if self.num_advice_queries else 1
        return max(3, factors) + 2

    def minimum_rows(self):
        return self.blinding_factors() + 3


class GraphEvaluator:
    """Expression -> straight-line `Calculation` program with common sub-expressions shared."""

    def __init__(self):
        self.constants = [0, 1, 2]
        self.rotations = []
        self.calculations = []
        self._seen = {}

    def add_constant(self, v):
        if v not in self.constants:
            self.constants.append(v)
        return ev.vs(ev.VS_CONSTANT, self.constants.index(v))

    def add_rotation(self, r):
        if r not in self.rotations:
            self.rotations.append(r)
        return self.rotations.index(r)

    def _calc(self, op, a, b=None, challenge=0, power=0):
        key = (op, (a.kind, a.index, a.rot), (b.kind, b.index, b.rot) if b is not None else None, challenge, power)
        if key not in self._seen:
            self.calculations.append(ev.calc(op, a, b, challenge, power))
            self._seen[key] = len(self.calculations) - 1
        return ev.vs(ev.VS_INTERMEDIATE, self._seen[key])

    def add_calculation(self, c):
        """an already-built `Calculation` becomes an intermediate"""
        return self._calc(c.op, c.a, c.b, c.challenge, c.power)

    def add_expression(self, e):
        if isinstance(e, Constant):
            return self.add_constant(e.v)
        if isinstance(e, Query):
            # the reference wraps every query in a Store calculation (evaluation.rs add_expression); the device
            # interpreter takes column operands directly, which saves one intermediate round trip per query
            return ev.vs(e.kind, e.column, self.add_rotation(e.rotation))
        if isinstance(e, Negated):
            return self._calc(ev.CALC_NEGATE, self.add_expression(e.e))
        if isinstance(e, Sum):
            if isinstance(e.b, Negated):
                return self._calc(ev.CALC_SUB, self.add_expression(e.a), self.add_expression(e.b.e))
            return self._calc(ev.CALC_ADD, self.add_expression(e.a), self.add_expression(e.b))
        if isinstance(e, Product):
            return self._calc(ev.CALC_MUL, self.add_expression(e.a), self.add_expression(e.b))
        if isinstance(e, Scaled):
            return self._calc(ev.CALC_MUL, self.add_expression(e.e), self.add_constant(e.c))
        raise TypeError(e)


def compile_gates(cs):
    """Evaluator::new (plonk/evaluation.rs:298-330): every polynomial of every gate becomes one value part"""
    g, parts, _, _ = compile_evaluator(cs)
    return g, parts


def _evaluate_lc(g, expressions):
    """theta-compression of an expression list (evaluation.rs:343-353)"""
    parts = [g.add_expression(e) for e in expressions]
    lc = parts[0]
    for part in parts[1:]:
        lc = g._calc(ev.CALC_LC_THETA, lc, part)
    return lc


def compile_evaluator(cs):
    """Evaluator::new (plonk/evaluation.rs:307-575): gates -> value parts; per lookup (table + beta, per input set the
    product of the phi_i = f_i + beta and the sum of the all-but-one products); per shuffle group the two running
    products with the challenges beta^(i+1).  The per-argument results are `Calculation`s evaluated after the shared
    program, exactly the `lookup_results` / `shuffle_results` the device interpreter expects."""
    g = GraphEvaluator()
    parts = [g.add_expression(p) for _, polys in cs.gates for p in polys]
    one = g.add_constant(1)
    lookups = []
    for _, table, sets in cs.lookups:
        table_calc = ev.calc(ev.CALC_ADD_CHALLENGE, _evaluate_lc(g, table), None, ev.CHALLENGE_BETA)
        phis = [[g._calc(ev.CALC_ADD_CHALLENGE, _evaluate_lc(g, inputs), None, ev.CHALLENGE_BETA) for inputs in st]
                for st in sets]
        prods, sums = [], []
        for phi in phis:
            prod = phi[0]
            for p_ in phi[1:]:
                prod = g._calc(ev.CALC_MUL, prod, p_)
            prods.append(ev.calc(ev.CALC_STORE, prod))
        for phi in phis:
            if len(phi) > 1:
                terms = []
                for i in range(len(phi)):
                    rest = [v for j, v in enumerate(phi) if j != i]
                    acc = rest[0]
                    for v in rest[1:]:
                        acc = g._calc(ev.CALC_MUL, acc, v)
                    terms.append(acc)
                total = terms[0]
                for v in terms[1:]:
                    total = g._calc(ev.CALC_ADD, total, v)
                sums.append(ev.calc(ev.CALC_STORE, total))
            else:
                sums.append(ev.calc(ev.CALC_STORE, one))
        lookups.append((table_calc, prods, sums))
    shuffles = []
    for group in cs.shuffles:
        ins = [_evaluate_lc(g, inp) for _, inp, _ in group]
        shs = [_evaluate_lc(g, shf) for _, _, shf in group]

        def running(vals):
            c = ev.calc(ev.CALC_ADD_CHALLENGE, vals[0], None, ev.CHALLENGE_BETA)
            for i, part in enumerate(vals[1:], start=1):
                c = ev.calc(ev.CALC_LC_CHALLENGE, part, g.add_calculation(c), ev.CHALLENGE_BETA, i + 1)
            return c

        shuffles.append((running(ins), running(shs)))
    if not g.rotations:
        g.add_rotation(0)
    return g, parts, lookups, shuffles


def compile_compress(expressions):
    """evaluate_with_theta (plonk/evaluation.rs:2330-2398) as a program: the value parts, Horner-folded by the
    interpreter with y := theta on the n-point Lagrange domain (extended_k := k)"""
    g = GraphEvaluator()
    parts = [g.add_expression(e) for e in expressions]
    if not g.rotations:
        g.add_rotation(0)
    return g, parts
