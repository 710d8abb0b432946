"""The slice of `plonk::circuit` the prover's hot path needs on the host: column allocation, queries, custom gates,
the equality (permutation) argument, and the flattening of gate expressions into the `Calculation` program that
h2_dev_evaluate_h interprets.

  ConstraintSystem   plonk/circuit.rs:1283-1956  (advice_column, fixed_column, enable_equality, create_gate,
                                                  query_*_index, degree :1881-1914, blinding_factors :1919-1944)
  Expression         plonk/circuit.rs:609-1060
  GraphEvaluator     plonk/evaluation.rs:298-560  (add_expression / add_calculation / add_constant / add_rotation)

Lookups and shuffles are not wired into the host prover here (the device interpreter supports them, see
evaluation.py); circuits are custom gates + copy constraints.
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


class Query(Expression):
    kind = None

    def __init__(self, column, rotation):
        self.column, self.rotation = column, rotation

    def degree(self):
        return 1


class Fixed(Query):
    kind = ev.VS_FIXED


class Advice(Query):
    kind = ev.VS_ADVICE


class Instance(Query):
    kind = ev.VS_INSTANCE


class Negated(Expression):
    def __init__(self, e):
        self.e = e

    def degree(self):
        return self.e.degree()


class Sum(Expression):
    def __init__(self, a, b):
        self.a, self.b = a, b

    def degree(self):
        return max(self.a.degree(), self.b.degree())


class Product(Expression):
    def __init__(self, a, b):
        self.a, self.b = a, b

    def degree(self):
        return self.a.degree() + self.b.degree()


class Scaled(Expression):
    def __init__(self, e, c):
        self.e, self.c = e, c

    def degree(self):
        return self.e.degree()


class ConstraintSystem:
    def __init__(self, name="circuit"):
        self.name = name
        self.num_advice = self.num_fixed = self.num_instance = 0
        self.advice_queries, self.fixed_queries, self.instance_queries = [], [], []
        self.num_advice_queries = []
        self.gates = []              # (name, [Expression])
        self.perm_columns = []       # ("advice" | "fixed" | "instance", index)
        self.minimum_degree = None

    # -- columns ------------------------------------------------------------------------------------------
    def advice_column(self):
        self.num_advice += 1
        self.num_advice_queries.append(0)
        return ("advice", self.num_advice - 1)

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

    # -- derived quantities -------------------------------------------------------------------------------
    def degree(self):
        d = 3  # permutation::Argument::required_degree (plonk/permutation.rs:42-69)
        for _, polys in self.gates:
            for p in polys:
                d = max(d, p.degree())
        return max(d, self.minimum_degree or 1)

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

    def _calc(self, op, a, b=None):
        key = (op, (a.kind, a.index, a.rot), (b.kind, b.index, b.rot) if b is not None else None)
        if key not in self._seen:
            self.calculations.append(ev.calc(op, a, b))
            self._seen[key] = len(self.calculations) - 1
        return ev.vs(ev.VS_INTERMEDIATE, self._seen[key])

    def add_expression(self, e):
        if isinstance(e, Constant):
            return self.add_constant(e.v)
        if isinstance(e, Query):
            return self._calc(ev.CALC_STORE, ev.vs(e.kind, e.column, self.add_rotation(e.rotation)))
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
    g = GraphEvaluator()
    parts = [g.add_expression(p) for _, polys in cs.gates for p in polys]
    if not g.rotations:
        g.add_rotation(0)
    return g, parts
