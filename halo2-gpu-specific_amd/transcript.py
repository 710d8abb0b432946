"""Blake2b transcript and the byte encodings of the proof stream -- transcript.rs:15-300
(`Blake2bWrite` / `Blake2bRead` with `Challenge255`).  Host logic on Python integers; points arrive from the
device as Jacobian Montgomery limbs and are normalised here (the reference's `to_affine` / `batch_normalize`).

Conventions that cannot be checked against pairing_bn256@30b052f without a Rust toolchain ("parity unpinned",
DESIGN.md): the compressed point is x little-endian with bit 7 of byte 31 carrying the parity of y and the
identity encoded as 32 zero bytes; challenges are `from_bytes_wide` = the 512-bit little-endian digest mod r.
"""
import hashlib

import numpy as np

R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
Q_MOD = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
_MONT_INV_R = pow(1 << 256, -1, R_MOD)
_MONT_INV_Q = pow(1 << 256, -1, Q_MOD)
_M64 = (1 << 64) - 1

BLAKE2B_PREFIX_CHALLENGE = b"\x00"
BLAKE2B_PREFIX_POINT = b"\x01"
BLAKE2B_PREFIX_SCALAR = b"\x02"


def limbs_to_int(l):
    return int(l[0]) | int(l[1]) << 64 | int(l[2]) << 128 | int(l[3]) << 192


def fr_from_mont_limbs(l):
    return limbs_to_int(l) * _MONT_INV_R % R_MOD


def fr_to_mont_limbs(v):
    m = (v << 256) % R_MOD
    return [(m >> (64 * i)) & _M64 for i in range(4)]


def jacobian_to_affine(xyz):
    """12 u64 (x, y, z Montgomery Fq limbs, Jacobian) -> (x, y) canonical integers, or None for the identity"""
    x, y, z = (limbs_to_int(xyz[4 * i:4 * i + 4]) * _MONT_INV_Q % Q_MOD for i in range(3))
    if z == 0:
        return None
    zi = pow(z, -1, Q_MOD)
    zi2 = zi * zi % Q_MOD
    return (x * zi2 % Q_MOD, y * zi2 % Q_MOD * zi % Q_MOD)


def jacobians_to_affine(rows):
    """jacobian_to_affine for a batch of points with ONE modular inversion (Montgomery's trick; an inversion is ~15 us of
    Python, the products ~0.3 us: the commitments of a phase are normalised together before they are hashed)"""
    raw = np.ascontiguousarray(rows, dtype=np.uint64).tobytes()       # one conversion instead of twelve scalars per point
    pts = [[int.from_bytes(raw[96 * j + 32 * i:96 * j + 32 * i + 32], "little") * _MONT_INV_Q % Q_MOD for i in range(3)]
           for j in range(len(raw) // 96)]
    prefix, acc = [], 1
    for _, _, z in pts:
        prefix.append(acc)
        if z:
            acc = acc * z % Q_MOD
    inv = pow(acc, -1, Q_MOD)
    out = [None] * len(pts)
    for i in range(len(pts) - 1, -1, -1):
        x, y, z = pts[i]
        if z == 0:
            continue
        zi = inv * prefix[i] % Q_MOD
        inv = inv * z % Q_MOD
        zi2 = zi * zi % Q_MOD
        out[i] = (x * zi2 % Q_MOD, y * zi2 % Q_MOD * zi % Q_MOD)
    return out


def g1_add_affine(P, Q):
    """P + Q on BN254 G1 for affine points as returned by jacobian_to_affine (None = the identity)"""
    if P is None:
        return Q
    if Q is None:
        return P
    (x1, y1), (x2, y2) = P, Q
    if x1 == x2:
        if (y1 + y2) % Q_MOD == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, Q_MOD) % Q_MOD
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, Q_MOD) % Q_MOD
    x3 = (lam * lam - x1 - x2) % Q_MOD
    return (x3, (lam * (x1 - x3) - y1) % Q_MOD)


def point_to_bytes(P):
    if P is None:
        return bytes(32)
    b = bytearray(P[0].to_bytes(32, "little"))
    b[31] |= (P[1] & 1) << 7
    return bytes(b)


class Blake2bWrite:
    """transcript.rs:152-226"""

    def __init__(self):
        self.state = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")
        self.writer = bytearray()

    def squeeze_challenge_scalar(self):
        self.state.update(BLAKE2B_PREFIX_CHALLENGE)
        return int.from_bytes(self.state.copy().digest(), "little") % R_MOD

    def common_point(self, P):
        if P is None:
            raise ValueError("cannot write points at infinity to the transcript")
        self.state.update(BLAKE2B_PREFIX_POINT)
        self.state.update(P[0].to_bytes(32, "little"))
        self.state.update(P[1].to_bytes(32, "little"))

    def common_scalar(self, v):
        self.state.update(BLAKE2B_PREFIX_SCALAR)
        self.state.update(v.to_bytes(32, "little"))

    def write_point(self, P):
        self.common_point(P)
        self.writer += point_to_bytes(P)

    def write_scalar(self, v):
        self.common_scalar(v)
        self.writer += v.to_bytes(32, "little")

    def finalize(self):
        return bytes(self.writer)
