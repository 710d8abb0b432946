"""On-disk formats on either side of the hot path.

  SRS file       Params::{write, read}  poly/commitment.rs:241-294
                 u32 k | n x 32 B g (compressed) | n x 32 B g_lagrange (compressed) | u32 len | additional_data
                 The reference decompresses with `from_bytes` under a rayon `parallelize`; here the 2 x n square
                 roots run on the device (h2_dev_points_decompress) and the tables never visit the host as points.
  circuit data   CircuitData::{write, read}  plonk.rs:126-204 + helpers.rs (constraint system, verifying-key commitments,
                 fixed columns, permutation mapping) -- see the section at the end of this file
  witness file   AssignWitnessCollection::{store_witness, fetch_witness}  helpers.rs:920-1015
                 u32 columns | column i at byte offset 4 + (i << (k + 5)): n x 32 B raw (Montgomery) Fr
                 -- consumed by create_proof_from_witness (plonk/prover.rs:916-1500)

The point encoding (x little-endian, y parity in bit 7 of byte 31, identity = zeros) is this build's convention:
pairing_bn256@30b052f is not available to check against ("parity unpinned").
"""
import struct

import numpy as np

from . import circuit
from ._lib import check
from .prover import Params
from .transcript import point_to_bytes


def params_write(device, params, path, additional_data=b""):
    """Params::write.  additional_data: the compressed [s]G2 of the setup (opaque here)."""
    torch = device.torch
    with open(path, "wb") as f:
        f.write(struct.pack("<I", params.k))
        for table in (params.g, params.g_lagrange):
            with torch.cuda.stream(device.tstream):
                out = torch.empty((params.n, 32), dtype=torch.uint8, device=device.dev)
            check(device.L.h2_dev_points_compress(table.data_ptr(), params.n, out.data_ptr(), device.stream),
                  "h2_dev_points_compress")
            with torch.cuda.stream(device.tstream):
                f.write(out.cpu().numpy().tobytes())
        f.write(struct.pack("<I", len(additional_data)))
        f.write(additional_data)


def params_read(device, path):
    """Params::read -> (Params with both tables resident on the device, additional_data)"""
    torch = device.torch
    with open(path, "rb") as f:
        (k,) = struct.unpack("<I", f.read(4))
        n = 1 << k
        tables = []
        for _ in range(2):
            raw = np.frombuffer(f.read(32 * n), dtype=np.uint8)
            if raw.size != 32 * n:
                raise IOError("truncated params file")
            with torch.cuda.stream(device.tstream):
                d_raw = torch.from_numpy(raw.copy()).to(device.dev)
                pts = torch.empty((n, 8), dtype=torch.int64, device=device.dev)
            check(device.L.h2_dev_points_decompress(d_raw.data_ptr(), n, pts.data_ptr(), device.stream),
                  "h2_dev_points_decompress")
            tables.append(pts)
        (alen,) = struct.unpack("<I", f.read(4))
        additional = f.read(alen)
        if len(additional) != alen:
            raise IOError("truncated params file")
    return Params(device, k, tables[0], tables[1]), additional


def witness_store(path, k, columns):
    """store_witness: columns = (n, 4) u64 arrays in the in-memory (Montgomery) representation"""
    n = 1 << k
    with open(path, "wb") as f:
        f.write(struct.pack("<I", len(columns)))
        for col in columns:
            col = np.ascontiguousarray(col, dtype=np.uint64)
            assert col.shape == (n, 4)
            f.write(col.tobytes())


def witness_fetch(path, k):
    """fetch_witness: the columns as read-only memory maps of the file (bundle size 2^(k+5) bytes)"""
    n = 1 << k
    with open(path, "rb") as f:
        (count,) = struct.unpack("<I", f.read(4))
    return [np.memmap(path, dtype=np.uint64, mode="r", offset=4 + (i << (k + 5)), shape=(n, 4)) for i in range(count)]


# ---- CircuitData (plonk.rs:126-204; helpers.rs:64-112,237-758,902-917) ---------------------------------------------
#   u32 j (quotient degree + 1 = cs.degree()) | u32 k | constraint system (write_cs, helpers.rs:406-456) |
#   fixed commitments, permutation commitments (32 B compressed each, plonk.rs:60-67) |
#   fixed columns: u32 count, each u32 n + n x 32 B raw (Montgomery) Fr (helpers.rs:182-199, 237-253) |
#   permutation mapping: u32 columns, u32 length per column, then (u32 column, u32 row) pairs (helpers.rs:114-179)
# Every integer is a little-endian u32; rotations are stored as `i32 as u32`; field constants as their canonical
# 32-byte little-endian representation.  Selectors are compiled away before a key is written (keygen.rs:357).
_ANY = {"advice": 0, "fixed": 1, "instance": 2}          # plonk/circuit.rs:79-86
_ANY_NAME = {v: k for k, v in _ANY.items()}
_E_CONSTANT, _E_FIXED, _E_ADVICE, _E_INSTANCE, _E_NEGATED, _E_SUM, _E_PRODUCT, _E_SCALED = range(8)   # helpers.rs:590-599


def _u32(v):
    return struct.pack("<I", v & 0xFFFFFFFF)


class _Reader:
    def __init__(self, buf):
        self.buf, self.pos = buf, 0

    def take(self, n):
        if self.pos + n > len(self.buf):
            raise IOError("truncated circuit data")
        out = self.buf[self.pos:self.pos + n]
        self.pos += n
        return out

    def u32(self):
        return struct.unpack("<I", self.take(4))[0]

    def i32(self):
        return struct.unpack("<i", self.take(4))[0]

    def fr(self):
        v = int.from_bytes(self.take(32), "little")
        if v >= circuit.R_MOD:
            raise IOError("non-canonical field element in circuit data")
        return v


def _expression_store(cs, e, out):
    """Expression::store (helpers.rs:687-757)"""
    if isinstance(e, circuit.Constant):
        out += [_u32(_E_CONSTANT), (e.v % circuit.R_MOD).to_bytes(32, "little")]
    elif isinstance(e, circuit.Query):
        code, kind = {circuit.Fixed: (_E_FIXED, "fixed"), circuit.Advice: (_E_ADVICE, "advice"),
                      circuit.Instance: (_E_INSTANCE, "instance")}[type(e)]
        out += [_u32(code), _u32(cs.get_any_query_index((kind, e.column), e.rotation)), _u32(e.column), _u32(e.rotation)]
    elif isinstance(e, circuit.Negated):
        out.append(_u32(_E_NEGATED))
        _expression_store(cs, e.e, out)
    elif isinstance(e, (circuit.Sum, circuit.Product)):
        out.append(_u32(_E_SUM if isinstance(e, circuit.Sum) else _E_PRODUCT))
        _expression_store(cs, e.a, out)
        _expression_store(cs, e.b, out)
    elif isinstance(e, circuit.Scaled):
        out.append(_u32(_E_SCALED))
        _expression_store(cs, e.e, out)
        out.append((e.c % circuit.R_MOD).to_bytes(32, "little"))
    else:
        raise TypeError("cannot serialise %r" % (e,))


def _expression_fetch(r):
    """Expression::fetch (helpers.rs:628-685)"""
    code = r.u32()
    if code == _E_CONSTANT:
        return circuit.Constant(r.fr())
    if code in (_E_FIXED, _E_ADVICE, _E_INSTANCE):
        r.u32()  # query_index: implied by the query lists
        column, rotation = r.u32(), r.i32()
        return {_E_FIXED: circuit.Fixed, _E_ADVICE: circuit.Advice, _E_INSTANCE: circuit.Instance}[code](column, rotation)
    if code == _E_NEGATED:
        return circuit.Negated(_expression_fetch(r))
    if code in (_E_SUM, _E_PRODUCT):
        a = _expression_fetch(r)
        b = _expression_fetch(r)
        return (circuit.Sum if code == _E_SUM else circuit.Product)(a, b)
    if code == _E_SCALED:
        e = _expression_fetch(r)
        return circuit.Scaled(e, r.fr())
    raise IOError("unknown expression code %d" % code)


def _expressions_store(cs, exprs, out):
    out.append(_u32(len(exprs)))
    for e in exprs:
        _expression_store(cs, e, out)


def _expressions_fetch(r):
    return [_expression_fetch(r) for _ in range(r.u32())]


def _queried_cells(e, cells):
    """the (column, rotation) pairs a gate polynomial touches, in first-use order (Gate::queried_cells)"""
    if isinstance(e, circuit.Query):
        cell = (e.name, e.column, e.rotation)
        if cell not in cells:
            cells.append(cell)
    for child in ("e", "a", "b"):
        if hasattr(e, child):
            _queried_cells(getattr(e, child), cells)


def cs_store(cs):
    """write_cs (helpers.rs:406-456) -> bytes"""
    out = [_u32(cs.num_advice), _u32(cs.num_instance), _u32(0), _u32(cs.num_fixed), _u32(len(cs.num_advice_queries))]
    out += [_u32(v) for v in cs.num_advice_queries]
    out += [_u32(0), _u32(0)]                                   # selector_map, constants: no selectors / constant columns
    for queries in (cs.advice_queries, cs.instance_queries, cs.fixed_queries):
        out.append(_u32(len(queries)))
        for column, rotation in queries:
            out += [_u32(column), _u32(rotation)]
    out.append(_u32(len(cs.perm_columns)))
    for kind, index in cs.perm_columns:
        out += [_u32(index), _u32(_ANY[kind])]
    out.append(_u32(len(cs.lookups)))
    for _, table, sets in cs.lookups:
        out.append(_u32(len(sets)))
        for st in sets:
            out.append(_u32(len(st)))
            for inputs in st:
                _expressions_store(cs, inputs, out)
        _expressions_store(cs, table, out)
    out.append(_u32(len(cs.shuffles)))
    for group in cs.shuffles:
        out.append(_u32(len(group)))
        for _, inputs, shuffle in group:
            _expressions_store(cs, inputs, out)
            _expressions_store(cs, shuffle, out)
    out.append(_u32(len(cs.range_checks)))                      # range_check arguments (helpers.rs:444-451)
    for origin, sort, vmin, vmax, step in cs.range_checks:
        out += [_u32(origin), _u32(sort), _u32(vmin), _u32(vmax), _u32(step)]
    out.append(_u32(0))                                         # named_advices
    out.append(_u32(len(cs.gates)))
    for _, polys in cs.gates:
        _expressions_store(cs, polys, out)
        cells = []
        for p in polys:
            _queried_cells(p, cells)
        out.append(_u32(len(cells)))
        for kind, column, rotation in cells:
            out += [_u32(column), _u32(_ANY[kind]), _u32(rotation)]
    return b"".join(out)


def cs_fetch(r, name="circuit"):
    """read_cs (helpers.rs:458-561) -> ConstraintSystem; refuses what this prover does not implement (selectors that
    were not compiled away, constant columns)"""
    cs = circuit.ConstraintSystem(name)
    cs.num_advice, cs.num_instance = r.u32(), r.u32()
    if r.u32():
        raise IOError("circuit data with live selectors")
    cs.num_fixed = r.u32()
    cs.num_advice_queries = [r.u32() for _ in range(r.u32())]
    selector_map = [r.u32() for _ in range(r.u32())]
    constants = [r.u32() for _ in range(r.u32())]
    del selector_map, constants                                 # keygen-time information only
    lists = []
    for _ in range(3):
        lists.append([(r.u32(), r.i32()) for _ in range(r.u32())])
    cs.advice_queries, cs.instance_queries, cs.fixed_queries = lists
    for _ in range(r.u32()):
        index, kind = r.u32(), r.u32()
        cs.perm_columns.append((_ANY_NAME[kind], index))
    for _ in range(r.u32()):
        sets = [[_expressions_fetch(r) for _ in range(r.u32())] for _ in range(r.u32())]
        table = _expressions_fetch(r)
        cs.lookups.append(("", table, sets))
    for _ in range(r.u32()):
        group = []
        for _ in range(r.u32()):
            inputs = _expressions_fetch(r)
            group.append(("", inputs, _expressions_fetch(r)))
        cs.shuffles.append(group)
    for _ in range(r.u32()):                                    # range_check arguments (helpers.rs:520-536)
        cs.range_checks.append((r.u32(), r.u32(), r.u32(), r.u32(), r.u32()))
    for _ in range(r.u32()):                                    # named_advices: (String, u32)
        r.take(r.u32())
        r.u32()
    for _ in range(r.u32()):
        polys = _expressions_fetch(r)
        for _ in range(r.u32()):
            r.take(12)                                          # queried cells: recomputable from the polynomials
        cs.gates.append(("", polys))
    return cs


def circuit_data_write(path, device, params, pk):
    """CircuitData::write for a key made by prover.keygen"""
    cs, n = pk.cs, params.n
    with open(path, "wb") as f:
        f.write(_u32(cs.degree()) + _u32(params.k))
        f.write(cs_store(cs))
        for P in list(pk.fixed_commitments) + list(pk.perm_commitments):
            f.write(point_to_bytes(P))
        f.write(_u32(len(pk.fixed_values)))
        for t in pk.fixed_values:
            f.write(_u32(n))
            f.write(np.ascontiguousarray(device.download(t)).tobytes())
        map_col, map_row = pk.mapping
        f.write(_u32(len(map_col)))
        for col in map_col:
            f.write(_u32(len(col)))
        for mc, mr in zip(map_col, map_row):
            f.write(np.stack([mc, mr], axis=1).astype("<u4").tobytes())


def circuit_data_read(path, name="circuit"):
    """CircuitData::read -> dict(j, k, cs, fixed_commitments, perm_commitments (compressed bytes), fixed (Montgomery
    (n, 4) u64 columns), mapping (map_col, map_row)); `prover.keygen_from_info` turns it into a proving key
    (CircuitData::into_proving_key, plonk.rs:196-198)"""
    with open(path, "rb") as f:
        r = _Reader(f.read())
    j, k = r.u32(), r.u32()
    cs = cs_fetch(r, name)
    cs.set_minimum_degree(j)     # the domain the key was made for (a `set_minimum_degree` call is not part of the stream)
    try:
        fits = cs.degree() == j
    except AssertionError:
        fits = False
    if not fits:
        raise IOError("circuit data: the constraint system does not fit a domain of degree %d" % j)
    fixed_commitments = [bytes(r.take(32)) for _ in range(cs.num_fixed)]
    perm_commitments = [bytes(r.take(32)) for _ in range(len(cs.perm_columns))]
    fixed = []
    for _ in range(r.u32()):
        m = r.u32()
        fixed.append(np.frombuffer(r.take(32 * m), dtype=np.uint64).reshape(m, 4))
    lengths = [r.u32() for _ in range(r.u32())]
    map_col, map_row = [], []
    for m in lengths:
        pairs = np.frombuffer(r.take(8 * m), dtype="<u4").reshape(m, 2)
        map_col.append(np.ascontiguousarray(pairs[:, 0]))
        map_row.append(np.ascontiguousarray(pairs[:, 1]))
    return {"j": j, "k": k, "cs": cs, "fixed_commitments": fixed_commitments, "perm_commitments": perm_commitments,
            "fixed": fixed, "mapping": (map_col, map_row)}
